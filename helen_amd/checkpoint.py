"""Reading a HELEN checkpoint WITHOUT importing torch.

The `.pkl` the reference saves (models/ModelHander.py:109-133: `torch.save({model_state_dict, model_optimizer,
hidden_size, gru_layers, epochs}, path)`) is a pickle whose tensors are persistent ids into raw storages.  `import torch`
costs a `helen polish` run 1.3 s before anything can start -- a quarter of a chr20-sized run on an MI355X -- and the run
needs nothing of torch but these few arrays.  Both container formats are read here with the standard library and numpy:
  * the zip archive of torch >= 1.6: `<name>/data.pkl` + one little-endian storage per `<name>/data/<key>`;
  * the legacy stream of torch <= 1.5 (the reference's Docker image pins 1.4, Dockerfile/Dockerfile:1): magic number,
    protocol version and system info as three pickles, the object's pickle, the pickled list of storage keys, then per key
    an int64 element count followed by the elements.
Only what such a file needs is admitted by the unpickler: tensors, their storages, OrderedDict and plain containers;
anything else raises `UnsupportedCheckpoint`, and helen_amd.model_handler then reads the file through torch.load as
before.  -> {key: numpy array or plain Python value}, tensors as C-contiguous arrays of their own dtype.
"""
import io
import pickle
import struct
import zipfile
from collections import OrderedDict

import numpy as np

_MAGIC = 0x1950a86a20f9469cfc6c


class UnsupportedCheckpoint(Exception):
    pass


_STORAGE_DTYPES = {
    "FloatStorage": np.float32, "DoubleStorage": np.float64, "HalfStorage": np.float16, "LongStorage": np.int64,
    "IntStorage": np.int32, "ShortStorage": np.int16, "CharStorage": np.int8, "ByteStorage": np.uint8,
    "BoolStorage": np.bool_,
}


class _StorageType(object):
    def __init__(self, name):
        self.dtype = np.dtype(_STORAGE_DTYPES[name])


def _rebuild_tensor(storage, storage_offset, size, stride, *unused):
    """torch._utils._rebuild_tensor_v2: a strided view of a flat storage, made contiguous."""
    size, stride = tuple(size), tuple(stride)
    if len(size) == 0:
        if not 0 <= int(storage_offset) < len(storage):
            raise UnsupportedCheckpoint("a scalar of the checkpoint lies outside its storage: the file is truncated or corrupt")
        return storage[storage_offset:storage_offset + 1].reshape(()).copy()
    if any(n == 0 for n in size):
        return np.zeros(size, storage.dtype)
    # a truncated or corrupt file must not become an out-of-bounds view (garbage weights or a fault): the pickled
    # geometry is checked against the storage it claims to view
    storage_offset = int(storage_offset)
    if (len(size) != len(stride) or storage_offset < 0 or any(int(n) < 0 for n in size) or any(int(st) < 0 for st in stride)
            or storage_offset + sum((int(n) - 1) * int(st) for n, st in zip(size, stride)) >= len(storage)):
        raise UnsupportedCheckpoint("a tensor of the checkpoint (offset %s, size %s, stride %s) does not fit its storage of %d "
                                    "elements: the file is truncated or corrupt" % (storage_offset, size, stride, len(storage)))
    view = np.lib.stride_tricks.as_strided(storage[storage_offset:], shape=size,
                                           strides=tuple(s * storage.dtype.itemsize for s in stride))
    return np.ascontiguousarray(view)


def _rebuild_parameter(data, requires_grad, backward_hooks, *unused):
    return data


class _Unpickler(pickle.Unpickler):
    def __init__(self, file, load_storage):
        pickle.Unpickler.__init__(self, file, encoding="utf-8")
        self._load_storage = load_storage

    def find_class(self, module, name):
        if module == "torch._utils" and name in ("_rebuild_tensor_v2", "_rebuild_tensor"):
            return _rebuild_tensor
        if module == "torch._utils" and name == "_rebuild_parameter":
            return _rebuild_parameter
        if module == "torch" and name in _STORAGE_DTYPES:
            return _StorageType(name)
        if module == "collections" and name == "OrderedDict":
            return OrderedDict
        if module == "torch" and name == "Size":
            return tuple
        raise UnsupportedCheckpoint("the checkpoint refers to %s.%s" % (module, name))

    def persistent_load(self, pid):
        if not isinstance(pid, tuple) or not pid or pid[0] != "storage":
            raise UnsupportedCheckpoint("unknown persistent id %r" % (pid,))
        return self._load_storage(pid)


def _read_zip(path):
    with zipfile.ZipFile(path) as z:
        names = z.namelist()
        pkl = [n for n in names if n.endswith("/data.pkl") or n == "data.pkl"]
        if len(pkl) != 1:
            raise UnsupportedCheckpoint("not a torch zip archive (no data.pkl)")
        root = pkl[0][:-len("data.pkl")]
        order = [n for n in names if n == root + "byteorder"]
        if order and z.read(order[0]).strip() not in (b"little", b""):
            raise UnsupportedCheckpoint("big-endian checkpoint")
        cache = {}

        def load_storage(pid):
            _, stype, key, _location, numel = pid[:5]
            if key not in cache:
                raw = z.read(root + "data/" + str(key))
                arr = np.frombuffer(raw, dtype=stype.dtype)
                if arr.shape[0] < int(numel):
                    raise UnsupportedCheckpoint("storage %s is shorter than its element count" % key)
                cache[key] = arr
            return cache[key]
        return _Unpickler(io.BytesIO(z.read(pkl[0])), load_storage).load()


class _Lazy(object):
    """A storage of the legacy stream: its bytes follow the object's pickle, so views are resolved afterwards."""

    def __init__(self, key, dtype, numel):
        self.key, self.dtype, self.numel, self.data = key, dtype, int(numel), None


def _read_legacy(path):
    with open(path, "rb") as f:
        try:
            magic = pickle.load(f)
        except Exception:
            raise UnsupportedCheckpoint("not a pickle")
        if magic != _MAGIC:
            raise UnsupportedCheckpoint("not a torch checkpoint (magic number)")
        pickle.load(f)                  # protocol version
        info = pickle.load(f)           # system info
        if isinstance(info, dict) and info.get("little_endian") is False:
            raise UnsupportedCheckpoint("big-endian checkpoint")
        storages, pending = {}, []

        def load_storage(pid):
            _, stype, root_key, _location, numel = pid[:5]
            view = pid[5] if len(pid) > 5 else None
            if view is not None:
                raise UnsupportedCheckpoint("storage views of the legacy format")
            if root_key not in storages:
                storages[root_key] = _Lazy(root_key, stype.dtype, numel)
            return storages[root_key]

        def rebuild(storage, storage_offset, size, stride, *unused):
            pending.append([storage, storage_offset, tuple(size), tuple(stride)])
            return pending[-1]
        unp = _Unpickler(f, load_storage)
        original = unp.find_class

        def find_class(module, name):
            if module == "torch._utils" and name in ("_rebuild_tensor_v2", "_rebuild_tensor"):
                return rebuild
            return original(module, name)
        unp.find_class = find_class
        obj = unp.load()
        keys = pickle.load(f)
        for key in keys:
            if key not in storages:
                raise UnsupportedCheckpoint("storage %r is not referenced" % (key,))
            st = storages[key]
            (numel,) = struct.unpack("<q", f.read(8))
            raw = f.read(numel * st.dtype.itemsize)
            if len(raw) != numel * st.dtype.itemsize:
                raise UnsupportedCheckpoint("truncated storage %r" % (key,))
            st.data = np.frombuffer(raw, dtype=st.dtype)
        done = {}
        for p in pending:
            done[id(p)] = _rebuild_tensor(p[0].data, p[1], p[2], p[3])

        def resolve(x):
            if isinstance(x, list) and id(x) in done:
                return done[id(x)]
            if isinstance(x, OrderedDict):
                return OrderedDict((k, resolve(v)) for k, v in x.items())
            if isinstance(x, dict):
                return {k: resolve(v) for k, v in x.items()}
            if isinstance(x, list):
                return [resolve(v) for v in x]
            if isinstance(x, tuple):
                return tuple(resolve(v) for v in x)
            return x
        return resolve(obj)


def load(path):
    """The object a `torch.save` wrote to `path`, tensors as numpy arrays.  Raises UnsupportedCheckpoint for anything this
    reader does not take (the caller then goes through torch.load)."""
    try:
        if zipfile.is_zipfile(path):
            return _read_zip(path)
        return _read_legacy(path)
    except UnsupportedCheckpoint:
        raise
    except (pickle.UnpicklingError, EOFError, KeyError, ValueError, struct.error, zipfile.BadZipFile, AttributeError,
            TypeError, IndexError) as e:
        raise UnsupportedCheckpoint("%s: %s" % (type(e).__name__, e))


def load_simple_model_state(path):
    """-> (state dict {name: float32 array} with a leading `module.` stripped, hidden_size, gru_layers, epochs): what
    ModelHandler.load_simple_model reads from the reference's checkpoint (ModelHander.py:50-78), torch-free."""
    ck = load(path)
    if not isinstance(ck, dict) or "model_state_dict" not in ck:
        raise UnsupportedCheckpoint("no model_state_dict in the checkpoint")
    state = OrderedDict()
    for k, v in ck["model_state_dict"].items():
        if not isinstance(v, np.ndarray):
            raise UnsupportedCheckpoint("parameter %s is not a tensor" % k)
        state[k[7:] if k[0:7] == "module." else k] = np.ascontiguousarray(v, dtype=np.float32)
    return state, ck["hidden_size"], ck["gru_layers"], ck["epochs"]
