"""NativeEngine: the model replica `helen polish` itself runs on -- libhelen_hip.so through ctypes and NOTHING else.

helen_amd.engine.HelenEngine serves callers that live in torch (device tensors in, device tensors out: the operator
boundary, the tests, the benchmark).  The command-line path needs none of that: its images are numpy arrays in page-locked
slots, its labels go back into the same slots, and `import torch` alone costs 1.3 s of a run whose device time is 3.7 s
(a chr20-sized image set).  This class is the same C ABI without torch: helen_model_create, the slot pipeline
(helen_polish_slot_submit / _wait: upload, kernels and label download of consecutive slots overlap on three streams inside
the library), helen_host_alloc for the slots' page-locked memory.  The process then runs on the system's HIP runtime
(helen_amd._lib.load(with_torch=False)).
"""
import ctypes

import numpy as np

from . import _lib
from .options import ImageSizeOptions

_PAIRS = (
    ("enc_w_ih", "gru_encoder.weight_ih_l0"), ("enc_w_hh", "gru_encoder.weight_hh_l0"),
    ("enc_b_ih", "gru_encoder.bias_ih_l0"), ("enc_b_hh", "gru_encoder.bias_hh_l0"),
    ("dec_w_ih", "gru_decoder.weight_ih_l0"), ("dec_w_hh", "gru_decoder.weight_hh_l0"),
    ("dec_b_ih", "gru_decoder.bias_ih_l0"), ("dec_b_hh", "gru_decoder.bias_hh_l0"),
)


def _plain_numpy(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32))


def weights_struct(state_dict, as_numpy=_plain_numpy):
    """state_dict (reference names, TransducerModel.py:43-58; a leading `module.` left by
    DataParallel/DDP is stripped as ModelHander.py:70-75 does) -> (HelenWeightsC, keepalive)."""
    sd = {}
    for k, v in state_dict.items():
        sd[k[7:] if k.startswith("module.") else k] = v
    keep = []

    def arr(name):
        if name not in sd:
            raise KeyError("missing parameter '%s' in model state" % name)
        a = as_numpy(sd[name])
        keep.append(a)
        return a

    s = _lib.HelenWeightsC()
    s.features = arr("gru_encoder.weight_ih_l0").shape[1]
    s.hidden = arr("gru_encoder.weight_hh_l0").shape[1]
    s.n_base = arr("dense1_base.weight").shape[0]
    s.n_rle = arr("dense2_rle.weight").shape[0]
    fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))  # noqa: E731
    for field, name in _PAIRS:
        pair = getattr(s, field)
        pair[0] = fp(arr(name))
        pair[1] = fp(arr(name + "_reverse"))
    s.base_w = fp(arr("dense1_base.weight"))
    s.base_b = fp(arr("dense1_base.bias"))
    s.rle_w = fp(arr("dense2_rle.weight"))
    s.rle_b = fp(arr("dense2_rle.bias"))
    return s, keep


def device_count():
    """HIP devices visible to this process (0 = none), without torch."""
    n = ctypes.c_int(0)
    _lib.check(_lib.load(with_torch=False).helen_device_count(ctypes.byref(n)))
    return int(n.value)


class PinnedBlock(object):
    """`nbytes` of page-locked host memory of the HIP runtime (helen_host_alloc), seen as a uint8 numpy array."""

    def __init__(self, nbytes, device=0):
        self._lib = _lib.load(with_torch=False)
        self._ptr = ctypes.c_void_p()
        _lib.check(self._lib.helen_host_alloc(int(device), int(nbytes), ctypes.byref(self._ptr)))
        self.array = np.ctypeslib.as_array((ctypes.c_uint8 * int(nbytes)).from_address(self._ptr.value))

    def close(self):
        if self._ptr:
            self.array = None
            self._lib.helen_host_free(self._ptr)
            self._ptr = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeEngine(object):
    """One model replica bound to one GPU (one process per GPU, predict_gpu.py:223), torch-free."""

    def __init__(self, state_dict, device=0, max_windows=4096, precision="fp32"):
        self._lib = _lib.load(with_torch=False)
        self._handle = ctypes.c_void_p()
        self.device = int(device)
        self.max_windows = int(max_windows)
        prec = {"fp32": _lib.HELEN_PRECISION_FP32, "bf16": _lib.HELEN_PRECISION_BF16,
                "fp32x3": _lib.HELEN_PRECISION_FP32X3}[precision]
        s, keep = weights_struct(state_dict)
        _lib.check(self._lib.helen_model_create(ctypes.byref(s), self.device, self.max_windows, prec,
                                                ctypes.byref(self._handle)))
        del keep
        self.in_flight = 0

    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle:
            self._lib.helen_model_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_bytes(self):
        n = ctypes.c_size_t()
        _lib.check(self._lib.helen_model_device_bytes(self._handle, ctypes.byref(n)))
        return int(n.value)

    def submit(self, images, bases, rles):
        """One slot (at most max_windows windows) into the pipeline: images uint8 [n, 1000, 90], bases / rles uint8
        [n, 1000], all in page-locked memory; returns at once.  At most two slots in flight: wait() first."""
        n = int(images.shape[0])
        _lib.check(self._lib.helen_polish_slot_submit(self._handle, images.ctypes.data, n, bases.ctypes.data,
                                                      rles.ctypes.data, None))
        self.in_flight += 1

    def wait(self):
        """Blocks until the OLDEST submitted slot's labels are in its arrays."""
        _lib.check(self._lib.helen_polish_slot_wait(self._handle))
        self.in_flight -= 1

    def polish_host(self, images, out=None):
        """images uint8 [n, 1000, 90] in ANY host memory -> (bases, rles) uint8 [n, 1000]: helen_polish_host (synchronous,
        staged through the library's own page-locked mirrors when the caller's memory is pageable)."""
        images = np.ascontiguousarray(images, dtype=np.uint8)
        n = images.shape[0]
        L = ImageSizeOptions.SEQ_LENGTH
        if out is not None:
            bases, rles = out
            for a in (bases, rles):
                if a.dtype != np.uint8 or a.shape != (n, L) or not a.flags.c_contiguous:
                    raise ValueError("out arrays must be C-contiguous uint8 [n,1000]")
        else:
            bases, rles = np.empty((n, L), np.uint8), np.empty((n, L), np.uint8)
        _lib.check(self._lib.helen_polish_host(self._handle, images.ctypes.data, n, bases.ctypes.data, rles.ctypes.data, None))
        return bases, rles
