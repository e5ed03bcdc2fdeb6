"""CpuEngine: the TransducerGRU on the HOST behind the C ABI of libhelen_cpu.so (include/helen_cpu.h,
helen_amd/csrc/cpu_path.cpp) -- the engine of runs WITHOUT --gpu_mode, as the reference's ONNX Runtime session is
(models/predict_cpu.py:39-170).  The product's own code: nothing of oracle/.  The MI355X path never comes here:
HelenEngine has no fallback, and `--gpu_mode` without a GPU still fails loudly."""
import ctypes
import os

import numpy as np

from .options import ImageSizeOptions, TrainOptions

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libhelen_cpu.so")
EXPORTS = ("helen_cpu_abi_version", "helen_cpu_last_error", "helen_cpu_polish_batch", "helen_cpu_chunk_forward")
_lib = None


def load():
    """libhelen_cpu.so, built on demand like the other two libraries; ImportError when it cannot be had."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        from ._lib import _try_build
        _try_build("libhelen_cpu.so")
    if not os.path.exists(LIB_PATH):
        raise ImportError("helen_amd: %s not found; build it with `make -C helen_amd/csrc libhelen_cpu.so`" % LIB_PATH)
    from ._lib import HelenWeightsC
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.helen_cpu_abi_version.restype = ci
    lib.helen_cpu_abi_version.argtypes = []
    lib.helen_cpu_last_error.restype = ctypes.c_char_p
    lib.helen_cpu_last_error.argtypes = []
    lib.helen_cpu_polish_batch.restype = ci
    lib.helen_cpu_polish_batch.argtypes = [ctypes.POINTER(HelenWeightsC), vp, ci, vp, vp, vp, vp, ci]
    lib.helen_cpu_chunk_forward.restype = ci
    lib.helen_cpu_chunk_forward.argtypes = [ctypes.POINTER(HelenWeightsC), vp, vp, ci, ci, vp, vp, vp, ci]
    if lib.helen_cpu_abi_version() != 1:
        raise ImportError("libhelen_cpu.so ABI %d != binding ABI 1; rebuild" % lib.helen_cpu_abi_version())
    _lib = lib
    return lib


class CpuEngine(object):
    """One model replica on the host; `threads` OpenMP threads per call (threads_per_caller,
    CallConsensusInterface.py:131).  The surface predict() uses of HelenEngine: polish_host, chunk_forward, close."""

    device_bytes = 0

    def __init__(self, state_dict, threads=0):
        from .native_engine import weights_struct
        self._lib = load()
        self.threads = int(threads)
        self._weights, self._keep = weights_struct(state_dict)

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self._lib.helen_cpu_last_error().decode("utf-8", "replace"))

    def polish_host(self, images, out=None, want_acc=False):
        """images uint8 [n, 1000, 90] -> (bases, rles) uint8 [n, 1000] [, acc_base f32 [n,1000,5], acc_rle f32 [n,1000,11]]."""
        images = np.ascontiguousarray(images, dtype=np.uint8)
        n = images.shape[0]
        L = ImageSizeOptions.SEQ_LENGTH
        assert tuple(images.shape[1:]) == (L, ImageSizeOptions.IMAGE_HEIGHT)
        if out is not None:
            bases, rles = out
            for a in (bases, rles):
                if a.dtype != np.uint8 or a.shape != (n, L) or not a.flags.c_contiguous:
                    raise ValueError("out arrays must be C-contiguous uint8 [n,1000]")
        else:
            bases, rles = np.empty((n, L), np.uint8), np.empty((n, L), np.uint8)
        acc_b = acc_r = None
        if want_acc:
            acc_b = np.empty((n, L, ImageSizeOptions.TOTAL_BASE_LABELS), np.float32)
            acc_r = np.empty((n, L, ImageSizeOptions.TOTAL_RLE_LABELS), np.float32)
        if n:
            self._check(self._lib.helen_cpu_polish_batch(
                ctypes.byref(self._weights), images.ctypes.data, n, bases.ctypes.data, rles.ctypes.data,
                None if acc_b is None else acc_b.ctypes.data, None if acc_r is None else acc_r.ctypes.data, self.threads))
        return (bases, rles, acc_b, acc_r) if want_acc else (bases, rles)

    def chunk_forward(self, x, hidden):
        """TransducerGRU.forward: x f32 [B,T,90], hidden f32 [B,2,128] (numpy or CPU tensors) -> numpy (base, rle, hidden)."""
        x = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
        hidden = np.ascontiguousarray(np.asarray(hidden, dtype=np.float32))
        B, T, _ = x.shape
        base = np.empty((B, T, ImageSizeOptions.TOTAL_BASE_LABELS), np.float32)
        rle = np.empty((B, T, ImageSizeOptions.TOTAL_RLE_LABELS), np.float32)
        h_out = np.empty((B, 2, TrainOptions.HIDDEN_SIZE), np.float32)
        self._check(self._lib.helen_cpu_chunk_forward(ctypes.byref(self._weights), x.ctypes.data, hidden.ctypes.data, B, T,
                                                      base.ctypes.data, rle.ctypes.data, h_out.ctypes.data, self.threads))
        return base, rle, h_out

    def close(self):
        self._keep = None
