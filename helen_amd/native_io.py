"""ctypes binding of libhelen_io.so (helen_amd/csrc/io.cpp): batch-at-a-time HDF5 reader/writer.

Optional accelerator for the file I/O on either side of the hot path: when the library is not
built (no hdf5.h at build time) the pure-Python path in helen_amd/hdf5.py does the same job."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libhelen_io.so")
NAME_BYTES = 256
_lib = None
_tried = False


def load():
    """The library, or None if it is not available."""
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    if os.environ.get("HELEN_NO_NATIVE_IO"):
        return None
    if not os.path.exists(LIB_PATH):
        from ._lib import _try_build
        _try_build("libhelen_io.so")
    if not os.path.exists(LIB_PATH):
        return None
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError:
        return None
    vp = ctypes.c_void_p
    lib.helen_io_last_error.restype = ctypes.c_char_p
    lib.helen_io_list_images.restype = ctypes.c_int
    lib.helen_io_list_images.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t,
                                         ctypes.POINTER(ctypes.c_longlong)]
    lib.helen_io_read_labeled.restype = ctypes.c_int
    lib.helen_io_read_labeled.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, vp, vp, vp]
    lib.helen_io_emit_images.restype = ctypes.c_int
    lib.helen_io_emit_images.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, vp, vp, vp, vp]
    lib.helen_io_emit_image_windows.restype = ctypes.c_int
    lib.helen_io_emit_image_windows.argtypes = [ctypes.c_char_p, ctypes.c_int, vp, vp, vp, vp, vp, vp, vp]
    lib.helen_io_library_lock.restype = None
    lib.helen_io_library_lock.argtypes = []
    lib.helen_io_library_unlock.restype = None
    lib.helen_io_library_unlock.argtypes = []
    lib.helen_io_reader_counts.restype = None
    lib.helen_io_reader_counts.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
    lib.helen_io_close_readers.restype = None
    lib.helen_io_close_readers.argtypes = []
    lib.helen_io_read_images.restype = ctypes.c_int
    lib.helen_io_read_images.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, vp, vp, vp, vp]
    lib.helen_io_index_images.restype = ctypes.c_int
    lib.helen_io_index_images.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_int)]
    lib.helen_io_image_names.restype = ctypes.c_int
    lib.helen_io_image_names.argtypes = [ctypes.c_char_p, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_char_p,
                                         ctypes.c_size_t, ctypes.POINTER(ctypes.c_longlong)]
    lib.helen_io_read_image_range.restype = ctypes.c_int
    lib.helen_io_read_image_range.argtypes = [ctypes.c_char_p, ctypes.c_longlong, ctypes.c_int, vp, vp, vp, vp,
                                              ctypes.POINTER(ctypes.c_longlong)]
    lib.helen_io_read_image_runs.restype = ctypes.c_int
    lib.helen_io_read_image_runs.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_longlong),
                                             ctypes.POINTER(ctypes.c_int), ctypes.c_int, vp, vp, vp, vp,
                                             ctypes.POINTER(ctypes.c_longlong)]
    lib.helen_io_image_storage.restype = ctypes.c_int
    lib.helen_io_image_storage.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    lib.helen_io_forget_images.restype = None
    lib.helen_io_forget_images.argtypes = [ctypes.c_char_p]
    lib.helen_io_writer_open.restype = vp
    lib.helen_io_writer_open.argtypes = [ctypes.c_char_p]
    lib.helen_io_write_predictions.restype = ctypes.c_int
    lib.helen_io_write_predictions.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp, vp]
    lib.helen_io_write_predictions_sel.restype = ctypes.c_int
    lib.helen_io_write_predictions_sel.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp, vp, vp]
    lib.helen_io_writer_close.restype = ctypes.c_int
    lib.helen_io_writer_close.argtypes = [vp]
    lib.helen_io_list_regions.restype = ctypes.c_int
    lib.helen_io_list_regions.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_longlong),
                                          ctypes.c_char_p, vp, vp]
    lib.helen_io_region_sequence.restype = ctypes.c_longlong
    lib.helen_io_region_sequence.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p,
                                             ctypes.c_char_p, ctypes.c_longlong]
    lib.helen_io_decode_regions.restype = ctypes.c_longlong
    lib.helen_io_decode_regions.argtypes = [ctypes.c_int, vp, vp, vp, vp, vp, ctypes.c_int, vp, ctypes.c_longlong, vp]
    lib.helen_ssw_join_batch.restype = ctypes.c_int
    lib.helen_ssw_join_batch.argtypes = [ctypes.c_int, vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, vp]
    lib.helen_ssw_fast_path.restype = ctypes.c_int
    lib.helen_ssw_fast_path.argtypes = [ctypes.c_int]
    lib.helen_ssw_fast_path_counts.restype = None
    lib.helen_ssw_fast_path_counts.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong)]
    lib.helen_ssw_fast_path_after_forward.restype = ctypes.c_longlong
    lib.helen_ssw_fast_path_after_forward.argtypes = []
    lib.helen_ssw_align.restype = ctypes.c_int
    lib.helen_ssw_align.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, ctypes.c_int]
    _lib = lib
    return lib


def available():
    return load() is not None


def _err(lib):
    return lib.helen_io_last_error().decode("utf-8", "replace")


def list_images(path):
    """Names under `images/` in name order, or None if the file has no such group."""
    lib = load()
    cap = 1 << 20
    while True:
        buf = ctypes.create_string_buffer(cap)
        n = ctypes.c_longlong()
        rc = lib.helen_io_list_images(os.fsencode(path), buf, cap, ctypes.byref(n))
        if rc == -2:
            cap = int(n.value) + 16
            continue
        if rc == 1:
            return None
        if rc != 0:
            raise IOError(_err(lib))
        return buf.raw.split(b"\0", 1)[0].decode().split("\n")[:n.value] if n.value else []


def index_images(path):
    """(number of images, read through libhdf5?) of one file, or None if it has no `images` group: the file's images in
    name order are addressed by position from here on (read_image_runs)."""
    lib = load()
    n, through = ctypes.c_longlong(), ctypes.c_int()
    rc = lib.helen_io_index_images(os.fsencode(path), ctypes.byref(n), ctypes.byref(through))
    if rc == 1:
        return None
    if rc != 0:
        raise IOError(_err(lib))
    return int(n.value), bool(through.value)


def image_names(path, first, count):
    """Names of images [first, first + count) of the file's index."""
    lib = load()
    cap = 64 * max(1, count) + 64
    while True:
        buf = ctypes.create_string_buffer(cap)
        need = ctypes.c_longlong()
        rc = lib.helen_io_image_names(os.fsencode(path), first, count, buf, cap, ctypes.byref(need))
        if rc == -2:
            cap = int(need.value) + 16
            continue
        if rc != 0:
            raise IOError(_err(lib))
        return buf.value.decode().split("\n")[:count] if count else []


def _raise_reader_error(lib):
    msg = _err(lib)
    if msg.startswith("IMAGE SIZE ERROR") or "contig name longer than" in msg:
        raise ValueError(msg)
    raise IOError(msg)


def read_image_runs(runs, threads, images, positions, meta, contigs):
    """runs = [(path, first, count), ...] read into consecutive rows of the arrays (as read_images) by `threads` native
    threads; the GIL is released for the duration.  Returns how many of the images libhdf5 had to read."""
    lib = load()
    n = len(runs)
    paths = (ctypes.c_char_p * n)(*[os.fsencode(r[0]) for r in runs])
    firsts = (ctypes.c_longlong * n)(*[int(r[1]) for r in runs])
    counts = (ctypes.c_int * n)(*[int(r[2]) for r in runs])
    through = ctypes.c_longlong()
    rc = lib.helen_io_read_image_runs(n, paths, firsts, counts, int(threads), images.ctypes.data, positions.ctypes.data,
                                      meta.ctypes.data, contigs.ctypes.data, ctypes.byref(through))
    if rc != 0:
        _raise_reader_error(lib)
    return int(through.value)


def image_storage(path):
    """How the images of a file are stored, judged by its first image: one of "contiguous", "chunked", "deflate" (read by
    the direct scanner; compact counts as contiguous) or "libhdf5" (the scanner does not take the file); None without
    images.  The classes are what helen_amd.host_plan has reader rates for."""
    lib = load()
    out = (ctypes.c_int * 4)()
    rc = lib.helen_io_image_storage(os.fsencode(path), out)
    if rc == 1:
        return None
    if rc != 0:
        raise IOError(_err(lib))
    if out[0] == 1:
        return "libhdf5"
    if out[3]:
        return "deflate"
    return "chunked" if out[1] == 2 else "contiguous"


def forget_images(path):
    """The reader has moved past `path`: drop its index and mapping (the unmap runs in the calling thread)."""
    lib = load()
    if lib is not None:
        lib.helen_io_forget_images(os.fsencode(path))


def emit_images(path, contig, starts, chunks, lengths, images):
    """A synthetic image file through the direct emitter (benchmark inputs; see helen_amd.synthetic)."""
    lib = load()
    starts = np.ascontiguousarray(starts, np.int64)
    chunks = np.ascontiguousarray(chunks, np.int64)
    lengths = np.ascontiguousarray(lengths, np.int32)
    images = np.ascontiguousarray(images, np.uint8)
    rc = lib.helen_io_emit_images(os.fsencode(path), int(starts.shape[0]), contig.encode(), starts.ctypes.data,
                                  chunks.ctypes.data, lengths.ctypes.data, images.ctypes.data)
    if rc != 0:
        raise IOError(_err(lib))


def emit_image_windows(path, contigs, starts, ends, chunks, lengths, images, positions):
    """Image files of a simulated assembly through the direct emitter (helen_amd.synthetic.write_assembly_dir): per-window
    contig names (list of str), contig_start / contig_end / feature_chunk_idx, stored rows, images and position rows."""
    lib = load()
    packed = pack_contigs(contigs)
    starts, ends, chunks = (np.ascontiguousarray(a, np.int64) for a in (starts, ends, chunks))
    lengths = np.ascontiguousarray(lengths, np.int32)
    images = np.ascontiguousarray(images, np.uint8)
    positions = np.ascontiguousarray(positions, np.int64)
    rc = lib.helen_io_emit_image_windows(os.fsencode(path), int(starts.shape[0]), packed.ctypes.data, starts.ctypes.data,
                                         ends.ctypes.data, chunks.ctypes.data, lengths.ctypes.data, images.ctypes.data,
                                         positions.ctypes.data)
    if rc != 0:
        raise IOError(_err(lib))


class library_lock(object):
    """Context manager: the lock libhelen_io.so takes around every call it makes into libhdf5 (io.cpp g_library_mutex).
    A thread that uses libhdf5 through another binding (helen_amd.hdf5) beside the native reader / writer holds it for
    the duration; without libhelen_io.so nothing native uses the library and the lock is a no-op."""

    def __enter__(self):
        self._lib = load()
        if self._lib is not None:
            self._lib.helen_io_library_lock()
        return self

    def __exit__(self, *exc):
        if self._lib is not None:
            self._lib.helen_io_library_unlock()


def close_readers():
    """Drop every cached read handle and mapping of this process (helen_io_close_readers).  The caches check a
    path's identity (device, inode, size, mtime) on every hit, so this is tidiness, not correctness: predict and
    perform_stitch call it when they start."""
    lib = load()
    if lib is not None:
        lib.helen_io_close_readers()


def reader_counts():
    """(images read by the direct scanner, images read through libhdf5) in this process."""
    out = (ctypes.c_longlong * 2)()
    load().helen_io_reader_counts(out)
    return int(out[0]), int(out[1])


def read_images(path, names, images, positions, meta, contigs):
    """Fill images u8 [n,1000,90], positions i64 [n,1000,3], meta i64 [n,3], contigs u8 [n,256]
    (all C-contiguous views, e.g. into shared memory) with the `names` of one file."""
    lib = load()
    n = len(names)
    rc = lib.helen_io_read_images(os.fsencode(path), "\n".join(names).encode(), n,
                                  images.ctypes.data, positions.ctypes.data, meta.ctypes.data,
                                  contigs.ctypes.data)
    if rc != 0:
        msg = _err(lib)
        if msg.startswith("IMAGE SIZE ERROR") or "contig name longer than" in msg:
            raise ValueError(msg)
        raise IOError(msg)


def read_labeled(path, names, images, label_base, label_rle):
    """Fill images u8 [n,1000,90] and label_base / label_rle u8 [n,1000] with the `names` of one file (the loader of
    the evaluation path: nothing is padded, any other shape is the reader's IMAGE SIZE ERROR)."""
    lib = load()
    rc = lib.helen_io_read_labeled(os.fsencode(path), "\n".join(names).encode(), len(names), images.ctypes.data,
                                   label_base.ctypes.data, label_rle.ctypes.data)
    if rc != 0:
        msg = _err(lib)
        if msg.startswith("IMAGE SIZE ERROR"):
            raise ValueError(msg)
        raise IOError(msg)


def contig_names(contigs):
    """u8 [n,NAME_BYTES] NUL-terminated -> list of str."""
    return [bytes(row).split(b"\0", 1)[0].decode() for row in np.asarray(contigs)]


def pack_contigs(names, out=None):
    arr = out if out is not None else np.zeros((len(names), NAME_BYTES), np.uint8)
    arr[:] = 0
    for i, s in enumerate(names):
        b = s.encode()
        if len(b) > NAME_BYTES - 1:      # never cut: two contigs sharing a prefix would merge into one group
            raise ValueError("contig name longer than %d bytes: %r" % (NAME_BYTES - 1, s))
        arr[i, :len(b)] = np.frombuffer(b, np.uint8)
    return arr


class Writer(object):
    """Prediction file writer (the native side of helen_amd.data_store.DataStore)."""

    def __init__(self, path):
        self._lib = load()
        self._h = self._lib.helen_io_writer_open(os.fsencode(path))
        if not self._h:
            raise IOError(_err(self._lib))

    def write(self, contigs, meta, positions, bases, rles, sel=None):
        """Write the batch rows (all, or those listed in `sel`)."""
        n = int(meta.shape[0])
        contigs = np.ascontiguousarray(contigs, np.uint8)
        meta = np.ascontiguousarray(meta, np.int64)
        positions = np.ascontiguousarray(positions, np.int64)
        bases = np.ascontiguousarray(bases, np.uint8)
        rles = np.ascontiguousarray(rles, np.uint8)
        if sel is not None:
            sel = np.ascontiguousarray(sel, np.int32)
            if sel.size and (sel.min() < 0 or sel.max() >= n):
                raise IndexError("row selection out of range")
            rc = self._lib.helen_io_write_predictions_sel(
                self._h, int(sel.size), sel.ctypes.data, contigs.ctypes.data, meta.ctypes.data,
                positions.ctypes.data, bases.ctypes.data, rles.ctypes.data)
        else:
            rc = self._lib.helen_io_write_predictions(self._h, n, contigs.ctypes.data, meta.ctypes.data,
                                                      positions.ctypes.data, bases.ctypes.data,
                                                      rles.ctypes.data)
        if rc != 0:
            raise IOError(_err(self._lib))

    def close(self):
        """The group structures and the superblock are written here: a failure (disk full) must not pass silently."""
        if self._h:
            h, self._h = self._h, None
            if self._lib.helen_io_writer_close(h) != 0:
                raise IOError(_err(self._lib))


def list_regions(path, contig):
    """[(region name, contig_start, contig_end)] of predictions/<contig> in name order; None if the file has no such
    contig."""
    lib = load()
    sizes = (ctypes.c_longlong * 2)()
    rc = lib.helen_io_list_regions(os.fsencode(path), contig.encode(), sizes, None, None, None)
    if rc == 1:
        return None
    if rc != 0:
        raise IOError(_err(lib))
    n = int(sizes[0])
    if n == 0:
        return []
    names = ctypes.create_string_buffer(int(sizes[1]) + 1)
    starts = np.zeros(n, np.int64)
    ends = np.zeros(n, np.int64)
    if lib.helen_io_list_regions(os.fsencode(path), contig.encode(), sizes, names, starts.ctypes.data,
                                 ends.ctypes.data) != 0:
        raise IOError(_err(lib))
    return list(zip(names.value.decode().split("\n"), starts.tolist(), ends.tolist()))


def region_sequence(path, contig, region, as_bytes=False):
    """Decoded sequence of one region of a prediction file (Stitch.small_chunk_stitch's inner loop)."""
    lib = load()
    cap = 1 << 16
    while True:
        buf = ctypes.create_string_buffer(cap)
        n = lib.helen_io_region_sequence(os.fsencode(path), contig.encode(), region.encode(), buf, cap)
        if n == -2:
            cap *= 4
            continue
        if n < 0:
            raise IOError(_err(lib))
        return buf.raw[:n] if as_bytes else buf.raw[:n].decode()


def fast_inflate():
    """Are deflated chunks inflated through libdeflate (helen_io_fast_inflate)?  False without the native library."""
    if not available():
        return False
    lib = load()
    return bool(lib.helen_io_fast_inflate()) if hasattr(lib, "helen_io_fast_inflate") else False


def decode_regions(first, rows, positions, bases, rles, threads=1):
    """Sequences of regions whose images are in memory (helen_io_decode_regions): region r = windows
    rows[first[r]:first[r + 1]] of positions int64 [*, 1000, 3] / bases, rles uint8 [*, 1000], listed in the string order
    of their chunk ids.  -> (blob bytes-like uint8 array, offsets int64 [n_regions + 1])"""
    lib = load()
    first = np.ascontiguousarray(first, np.int32)
    rows = np.ascontiguousarray(rows, np.int32)
    n = int(first.shape[0]) - 1
    for a, dt in ((positions, np.int64), (bases, np.uint8), (rles, np.uint8)):
        if a.dtype != dt or not a.flags.c_contiguous:
            raise ValueError("decode_regions wants C-contiguous int64 positions and uint8 labels")
    offsets = np.zeros(n + 1, np.int64)
    cap = int(rows.shape[0]) * 1000 * 10 + 16          # a row decodes to at most ten bases (run-length labels 0..10)
    # (a label above 10 cannot come out of the 11-class head; a foreign array that holds one gets the exact retry below)
    out = np.empty(min(cap, max(1 << 16, int(rows.shape[0]) * 1000 * 3)), np.uint8)
    while True:
        got = lib.helen_io_decode_regions(n, first.ctypes.data, rows.ctypes.data, positions.ctypes.data, bases.ctypes.data,
                                          rles.ctypes.data, int(threads), out.ctypes.data, int(out.shape[0]), offsets.ctypes.data)
        if got == -2:
            need = int(_err(lib).rsplit(" ", 1)[-1])
            out = np.empty(need + 16, np.uint8)
            continue
        if got < 0:
            raise IOError(_err(lib))
        return out[:got], offsets


def ssw_join_batch(blob, l_off, l_len, r_off, r_len, match, mismatch, gap_open, gap_extend, min_run):
    """`n` overlap alignments reduced to (score, pos_a, pos_b) each (helen_ssw_join_batch) -> int32 [n, 3]."""
    lib = load()
    n = int(len(l_off))
    out = np.zeros((n, 3), np.int32)
    if n == 0:
        return out
    l_off, r_off = np.ascontiguousarray(l_off, np.int64), np.ascontiguousarray(r_off, np.int64)
    l_len, r_len = np.ascontiguousarray(l_len, np.int32), np.ascontiguousarray(r_len, np.int32)
    buf = np.frombuffer(blob, np.uint8) if not isinstance(blob, np.ndarray) else blob
    rc = lib.helen_ssw_join_batch(n, buf.ctypes.data, l_off.ctypes.data, l_len.ctypes.data, r_off.ctypes.data,
                                  r_len.ctypes.data, match, mismatch, gap_open, gap_extend, min_run, out.ctypes.data)
    if rc != 0:
        raise IOError(_err(lib))
    return out


class Alignment(object):
    """What the reference's HELEN.Alignment exposes to Stitch (pybind_api.h:18-31)."""
    __slots__ = ("best_score", "reference_begin", "reference_end", "query_begin", "query_end",
                 "mismatches", "cigar_string")


def ssw_fast_path(enable=None):
    """The aligner's exact-overlap shortcut (include/helen_io.h): True / False switches it, None only asks -> the previous
    setting."""
    return bool(load().helen_ssw_fast_path(-1 if enable is None else int(bool(enable))))


def ssw_fast_path_counts():
    """(alignments the shortcut answered, alignments that went through the three passes) since the library was loaded."""
    hits, misses = ctypes.c_longlong(), ctypes.c_longlong()
    load().helen_ssw_fast_path_counts(ctypes.byref(hits), ctypes.byref(misses))
    return int(hits.value), int(misses.value)


def ssw_align(reference, query, match, mismatch, gap_open, gap_extend):
    """Aligner(match, mismatch, gap_open, gap_extend).SetReferenceSequence(reference);
    Align_cpp(query, Filter(), alignment, 0) -> Alignment (same numbers as the reference's SSW)."""
    lib = load()
    r = reference.encode() if isinstance(reference, str) else bytes(reference)
    q = query.encode() if isinstance(query, str) else bytes(query)
    out = (ctypes.c_int * 6)()
    cap = 16 * (len(r) + len(q)) + 64
    cig = ctypes.create_string_buffer(cap)
    rc = lib.helen_ssw_align(r, len(r), q, len(q), match, mismatch, gap_open, gap_extend, out, cig, cap)
    if rc < 0:
        raise RuntimeError("helen_ssw_align: internal inconsistency")
    a = Alignment()
    (a.best_score, a.reference_begin, a.reference_end, a.query_begin, a.query_end, a.mismatches) = list(out)
    a.cigar_string = cig.value.decode()
    return a
