"""SequenceDataset: MarginPolish image files -> batches of pileup windows.

Mirrors helen/modules/python/models/dataloader_predict.py:11-95 and the DataLoader around it
(models/predict_gpu.py:81-85): the index is (file, image name) in file order x HDF5 key order,
items are the reference's 7-tuple, short images are zero-padded to 1000 positions with (-1,-1,-1)
position rows, batching is sequential with a short last batch.  Differences in mechanism only:
files stay open per process (the reference re-opens per item), HDF5 is read through
helen_amd.hdf5 (ctypes on libhdf5), and batches can be produced by worker processes.
"""
import collections
import sys

import numpy as np

from . import hdf5, native_io
from .file_manager import get_file_paths_from_directory
from .options import ImageSizeOptions

Batch = collections.namedtuple(
    "Batch", "contig contig_start contig_end chunk_id images positions filenames")


class _FileCache(object):
    """Per-process cache of open read-only files."""

    def __init__(self, limit=64):
        self.limit = limit
        self.files = collections.OrderedDict()

    def get(self, path):
        f = self.files.get(path)
        if f is None:
            if len(self.files) >= self.limit:
                _, old = self.files.popitem(last=False)
                old.close()
            f = hdf5.File(path, "r")
            self.files[path] = f
        else:
            self.files.move_to_end(path)
        return f


_cache = _FileCache()


def read_item(hdf5_filepath, image_name):
    """One image with its bookkeeping, padded to SEQ_LENGTH (dataloader_predict.py:54-88)."""
    f = _cache.get(hdf5_filepath)
    base = "images/" + image_name + "/"
    # np.array2string(name.astype(np.str)).replace("'", '') (dataloader_predict.py:64): array2string prints the string
    # scalar as repr() does -- in double quotes when the name holds a single quote and no double quote
    contig = repr(str(f.read(base + "contig").reshape(-1)[0])).replace("'", "")
    contig_start = int(f.read(base + "contig_start", np.int64).reshape(-1)[0])
    contig_end = int(f.read(base + "contig_end", np.int64).reshape(-1)[0])
    chunk_id = int(f.read(base + "feature_chunk_idx", np.int64).reshape(-1)[0])
    image = f.read(base + "image", np.uint8)
    position = f.read(base + "position", np.int64)
    L, H = ImageSizeOptions.SEQ_LENGTH, ImageSizeOptions.IMAGE_HEIGHT
    if image.ndim != 2 or image.shape[1] != H or position.ndim != 2 or position.shape[1] != 3:
        raise ValueError("IMAGE SIZE ERROR: " + str(hdf5_filepath) + " " + str(image.shape))
    if image.shape[0] < L:
        need = L - image.shape[0]
        image = np.concatenate([image, np.zeros((need, H), np.uint8)], 0)
        position = np.concatenate([position, np.full((L - position.shape[0], 3), -1, np.int64)], 0)
    if image.shape[0] != L or position.shape[0] != L:
        raise ValueError("IMAGE SIZE ERROR: " + str(hdf5_filepath) + " " + str(image.shape))
    return contig, contig_start, contig_end, chunk_id, image, position, hdf5_filepath


def _collate(items):
    """Default-collate equivalent of torch's DataLoader for the 7-tuple (numpy instead of tensors)."""
    return Batch(
        contig=[it[0] for it in items],
        contig_start=np.array([it[1] for it in items], np.int64),
        contig_end=np.array([it[2] for it in items], np.int64),
        chunk_id=np.array([it[3] for it in items], np.int64),
        images=np.stack([it[4] for it in items]) if items else np.zeros((0, 1000, 90), np.uint8),
        positions=np.stack([it[5] for it in items]) if items else np.zeros((0, 1000, 3), np.int64),
        filenames=[it[6] for it in items])


def fill_batch(pairs, images, positions, meta, contigs):
    """Read the (file, image name) `pairs` into caller-provided arrays -- images u8 [n,1000,90],
    positions i64 [n,1000,3], meta i64 [n,3] (contig_start, contig_end, feature_chunk_idx), contigs
    u8 [n,128] -- with the reader's padding semantics.  Uses libhelen_io.so when it is built (one
    call per run of images from the same file), the ctypes/HDF5 path otherwise."""
    n = len(pairs)
    if native_io.available():
        i = 0
        while i < n:
            j = i
            while j < n and pairs[j][0] == pairs[i][0]:
                j += 1
            native_io.read_images(pairs[i][0], [name for _, name in pairs[i:j]], images[i:j],
                                  positions[i:j], meta[i:j], contigs[i:j])
            i = j
        return
    names = []
    for k, (path, name) in enumerate(pairs):
        contig, cs, ce, chunk, image, position, _ = read_item(path, name)
        images[k], positions[k] = image, position
        meta[k] = (cs, ce, chunk)
        names.append(contig)
    native_io.pack_contigs(names, contigs)


def _load_batch(pairs):
    n = len(pairs)
    L, H = ImageSizeOptions.SEQ_LENGTH, ImageSizeOptions.IMAGE_HEIGHT
    images = np.empty((n, L, H), np.uint8)
    positions = np.empty((n, L, 3), np.int64)
    meta = np.empty((n, 3), np.int64)
    contigs = np.zeros((n, native_io.NAME_BYTES), np.uint8)
    fill_batch(pairs, images, positions, meta, contigs)
    return Batch(contig=native_io.contig_names(contigs), contig_start=meta[:, 0].copy(),
                 contig_end=meta[:, 1].copy(), chunk_id=meta[:, 2].copy(), images=images,
                 positions=positions, filenames=[p for p, _ in pairs])


# ---- shared-memory slots: worker processes read straight into memory the parent hands to the GPU ----
class SharedSlot(object):
    """`cap` windows worth of reader output in one file-backed shared mapping (/dev/shm)."""

    def __init__(self, cap, path=None, create=True, prefix="helen_slot_"):
        import os
        import tempfile
        self.cap = int(cap)
        L, H = ImageSizeOptions.SEQ_LENGTH, ImageSizeOptions.IMAGE_HEIGHT
        self._sizes = (self.cap * L * H, self.cap * L * 3 * 8, self.cap * 3 * 8,
                       self.cap * native_io.NAME_BYTES, self.cap * L, self.cap * L)
        total = sum(self._sizes)
        if create:
            # RAM-backed when /dev/shm really has the pages (a container's default tmpfs is 64 MB, and touching a
            # page past a tmpfs' size is a SIGBUS, not an error code): the pages are RESERVED here with
            # posix_fallocate -- a sparse ftruncate would let every slot of every rank pass the same free-space
            # check -- and a refusal (ENOSPC) sends this slot to the temp directory instead.
            fd = path = None
            if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK):
                fd, path = tempfile.mkstemp(prefix=prefix, dir="/dev/shm")
                try:
                    os.posix_fallocate(fd, 0, total)
                except OSError:
                    os.close(fd)
                    _unlink_quietly(path)
                    fd = path = None
            if fd is None:
                fd, path = tempfile.mkstemp(prefix=prefix, dir=None)
                try:
                    os.posix_fallocate(fd, 0, total)
                except OSError:      # a file system without fallocate: plain (sparse) truncate
                    os.ftruncate(fd, total)
            os.close(fd)
            import atexit
            atexit.register(_unlink_quietly, path)      # also when the run dies with an exception
        self.path = path
        self.owner = create
        self._mm = np.memmap(path, dtype=np.uint8, mode="r+", shape=(total,))
        o0, o1, o2, o3, o4, o5 = np.cumsum((0,) + self._sizes[:5])
        self.images = self._mm[o0:o0 + self._sizes[0]].reshape(self.cap, L, H)
        self.positions = self._mm[o1:o1 + self._sizes[1]].view(np.int64).reshape(self.cap, L, 3)
        self.meta = self._mm[o2:o2 + self._sizes[2]].view(np.int64).reshape(self.cap, 3)
        self.contigs = self._mm[o3:o3 + self._sizes[3]].reshape(self.cap, native_io.NAME_BYTES)
        # label rows of the device call, for writer processes
        self.bases = self._mm[o4:o4 + self._sizes[4]].reshape(self.cap, L)
        self.rles = self._mm[o5:o5 + self._sizes[5]].reshape(self.cap, L)

    def base_address(self):
        """(address, bytes) of the whole mapping, for page-locking it in place."""
        return self._mm.ctypes.data, int(self._mm.nbytes)

    def close(self):
        import os
        self.images = self.positions = self.meta = self.contigs = self.bases = self.rles = None
        self._mm = None
        if self.owner and self.path and os.path.exists(self.path):
            os.unlink(self.path)


class PinnedSlot(object):
    """`cap` windows of reader output for reader THREADS of this process: the images and the label rows -- what the copy
    engines touch -- in page-locked host memory of the runtime (hipHostMalloc through torch; nothing to register, nothing
    to unregister, handed back to torch's host allocator at the end), positions / bounds / names in ordinary memory."""

    path = None          # (no file behind it: writer PROCESSES cannot attach)

    def __init__(self, cap, pin=True):
        import torch
        self.cap = int(cap)
        L, H = ImageSizeOptions.SEQ_LENGTH, ImageSizeOptions.IMAGE_HEIGHT
        pin = pin and torch.cuda.is_available()
        try:
            self.images_t = torch.empty((self.cap, L, H), dtype=torch.uint8, pin_memory=pin)
            self.bases_t = torch.empty((self.cap, L), dtype=torch.uint8, pin_memory=pin)
            self.rles_t = torch.empty((self.cap, L), dtype=torch.uint8, pin_memory=pin)
        except RuntimeError as e:       # page-locking refused (ulimit -l, container policy): ordinary memory, staged copies
            if not pin:
                raise
            import sys
            sys.stderr.write("INFO: SLOT NOT PAGE-LOCKED (" + str(e).splitlines()[0][:120] + "), USING STAGED COPIES.\n")
            self.images_t = torch.empty((self.cap, L, H), dtype=torch.uint8)
            self.bases_t = torch.empty((self.cap, L), dtype=torch.uint8)
            self.rles_t = torch.empty((self.cap, L), dtype=torch.uint8)
        self.images, self.bases, self.rles = self.images_t.numpy(), self.bases_t.numpy(), self.rles_t.numpy()
        self.positions = np.empty((self.cap, L, 3), np.int64)
        self.meta = np.empty((self.cap, 3), np.int64)
        self.contigs = np.zeros((self.cap, native_io.NAME_BYTES), np.uint8)

    def close(self):
        self.images = self.positions = self.meta = self.contigs = self.bases = self.rles = None
        self.images_t = self.bases_t = self.rles_t = None


class NativeSlot(object):
    """PinnedSlot without torch: images and label rows in ONE block of page-locked memory of the HIP runtime
    (helen_host_alloc through helen_amd.native_engine), positions / bounds / names in ordinary memory.  `pinned` says
    whether the block is page-locked (False: ordinary memory -- the host path, or a refused allocation -- and the device
    stage then goes through the library's staged, synchronous call)."""

    path = None

    def __init__(self, cap, device=0, pin=True):
        self.cap = int(cap)
        L, H = ImageSizeOptions.SEQ_LENGTH, ImageSizeOptions.IMAGE_HEIGHT
        n_img, n_lab = self.cap * L * H, self.cap * L
        self._block = None
        self.pinned = False
        if pin:
            try:
                from .native_engine import PinnedBlock
                self._block = PinnedBlock(n_img + 2 * n_lab + 8192, device if device is not None else 0)
                self.pinned = True
            except Exception as e:      # noqa: BLE001 -- page-locking refused (ulimit -l, container policy): staged copies
                import sys
                sys.stderr.write("INFO: SLOT NOT PAGE-LOCKED (" + str(e).splitlines()[0][:120] + "), USING STAGED COPIES.\n")
        raw = self._block.array if self._block is not None else np.empty(n_img + 2 * n_lab + 8192, np.uint8)
        lab0 = (n_img + 4095) // 4096 * 4096             # label rows on their own pages
        self.images = raw[:n_img].reshape(self.cap, L, H)
        self.bases = raw[lab0:lab0 + n_lab].reshape(self.cap, L)
        self.rles = raw[lab0 + n_lab:lab0 + 2 * n_lab].reshape(self.cap, L)
        self.positions = np.empty((self.cap, L, 3), np.int64)
        self.meta = np.empty((self.cap, 3), np.int64)
        self.contigs = np.zeros((self.cap, native_io.NAME_BYTES), np.uint8)

    def close(self):
        self.images = self.positions = self.meta = self.contigs = self.bases = self.rles = None
        if self._block is not None:
            self._block.close()
            self._block = None


def _unlink_quietly(path):
    import os
    try:
        os.unlink(path)
    except OSError:
        pass


_attached = {}


def attach_slot(path, cap):
    """This process's mapping of the slot another process created."""
    key = (path, cap)
    slot = _attached.get(key)
    if slot is None:
        if len(_attached) > 8:
            _attached.clear()
        slot = SharedSlot(cap, path=path, create=False)
        _attached[key] = slot
    return slot


class Filled(int):
    """len(pairs), plus how many of them libhdf5 had to read because the direct scanner declined the file."""
    through_library = 0


def fill_shared(path, cap, offset, pairs):
    """Worker entry: read `pairs` into slot `path` at window `offset`.  Returns len(pairs) (a `Filled`)."""
    slot = attach_slot(path, cap)
    n = len(pairs)
    before = native_io.reader_counts()[1] if native_io.available() else 0
    fill_batch(pairs, slot.images[offset:offset + n], slot.positions[offset:offset + n],
               slot.meta[offset:offset + n], slot.contigs[offset:offset + n])
    out = Filled(n)
    out.through_library = (native_io.reader_counts()[1] - before) if native_io.available() else n
    return out


class SequenceDataset(object):
    """dataloader_predict.py:18-52: every image of every file, files in the given order, a file's images in name order.
    `runs` = [(path, number of images, read through libhdf5?)] is that index by position (the native reader addresses an
    image as (file, position): helen_io_read_image_runs); `all_images` = the reference's list of (path, name) pairs,
    materialised when somebody asks for it."""

    def __init__(self, image_directory, file_list=None):
        if file_list is not None:
            hdf_files = list(file_list)
        else:
            hdf_files = get_file_paths_from_directory(image_directory)
        self.runs = []
        self._pairs = None
        if native_io.available():
            for path in hdf_files:
                got = native_io.index_images(path)
                if got is None:
                    sys.stderr.write("WARN: NO IMAGES FOUND IN FILE: " + path + "\n")
                    continue
                # (libhdf5's business if the scanner declines the file -- or, judged by the first image, its datasets)
                self.runs.append((path, got[0], got[1] or (got[0] > 0 and native_io.image_storage(path) == "libhdf5")))
            return
        pairs = []
        for path in hdf_files:
            with hdf5.File(path, "r") as f:
                names = f.keys("images") if "images" in f else None
            if names is None:
                sys.stderr.write("WARN: NO IMAGES FOUND IN FILE: " + path + "\n")
                continue
            self.runs.append((path, len(names), True))
            pairs.extend((path, name) for name in names)
        self._pairs = pairs

    @property
    def all_images(self):
        if self._pairs is None:
            pairs = []
            for path, n, _ in self.runs:
                pairs.extend((path, name) for name in native_io.image_names(path, 0, n))
            self._pairs = pairs
        return self._pairs

    @all_images.setter
    def all_images(self, pairs):
        self._pairs = pairs

    def call_runs(self, windows_per_call):
        """The index cut into device calls of `windows_per_call` consecutive images: [[(path, first, count), ...], ...]."""
        calls, cur, room = [], [], windows_per_call
        for path, n, _ in self.runs:
            first = 0
            while first < n:
                take = min(room, n - first)
                cur.append((path, first, take))
                first += take
                room -= take
                if room == 0:
                    calls.append(cur)
                    cur, room = [], windows_per_call
        if cur:
            calls.append(cur)
        return calls

    def __len__(self):
        return sum(n for _, n, _ in self.runs) if self._pairs is None else len(self._pairs)

    def __getitem__(self, index):
        path, name = self.all_images[index]
        return read_item(path, name)

    def num_batches(self, batch_size):
        return (len(self) + batch_size - 1) // batch_size

    def iter_batches(self, batch_size, num_workers=0, prefetch=4):
        """Sequential batches (shuffle=False, drop_last=False; predict_gpu.py:82-85).  With
        num_workers > 0 batches are read by a pool of worker processes, `prefetch` batches per
        worker in flight, and yielded in order."""
        groups = [self.all_images[i:i + batch_size] for i in range(0, len(self), batch_size)]
        if num_workers <= 0:
            for g in groups:
                yield _load_batch(g)
            return
        import concurrent.futures as cf
        import multiprocessing as mp
        ctx = mp.get_context("spawn")
        with cf.ProcessPoolExecutor(num_workers, mp_context=ctx) as pool:
            window = max(1, num_workers * prefetch)
            pending = collections.deque()
            it = iter(groups)
            for g in it:
                pending.append(pool.submit(_load_batch, g))
                if len(pending) >= window:
                    break
            while pending:
                batch = pending.popleft().result()
                nxt = next(it, None)
                if nxt is not None:
                    pending.append(pool.submit(_load_batch, nxt))
                yield batch
