"""HelenEngine: a device-resident TransducerGRU behind the C ABI.

PyTorch is plumbing here -- it owns device buffers and the stream; all compute is in
libhelen_hip.so (helen_amd/csrc).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .options import ImageSizeOptions, TrainOptions

from .native_engine import weights_struct as _weights_struct


def _as_numpy(v):
    if isinstance(v, torch.Tensor):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32))


def weights_struct(state_dict):
    """state_dict of torch tensors or arrays -> (HelenWeightsC, keepalive) (helen_amd.native_engine.weights_struct)."""
    return _weights_struct(state_dict, _as_numpy)


class HelenEngine(object):
    """One model replica bound to one GPU (one process per GPU, predict_gpu.py:223)."""

    def __init__(self, state_dict, device=0, max_windows=4096, precision="fp32"):
        self._lib = _lib.load()
        self._handle = ctypes.c_void_p()
        if not torch.cuda.is_available():
            raise RuntimeError("HelenEngine needs a GPU: torch.cuda.is_available() is False and "
                               "there is no CPU fallback")
        self.device = torch.device("cuda", int(device))
        self.max_windows = int(max_windows)
        prec = {"fp32": _lib.HELEN_PRECISION_FP32, "bf16": _lib.HELEN_PRECISION_BF16,
                "fp32x3": _lib.HELEN_PRECISION_FP32X3}[precision]
        s, keep = weights_struct(state_dict)
        _lib.check(self._lib.helen_model_create(ctypes.byref(s), self.device.index,
                                                self.max_windows, prec,
                                                ctypes.byref(self._handle)))
        del keep

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

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def polish(self, images, want_acc=False):
        """images: uint8 CUDA tensor [n, 1000, 90] -> (bases u8 [n,1000], rles u8 [n,1000]
        [, acc_base f32 [n,1000,5], acc_rle f32 [n,1000,11]]).  The per-batch body of the
        reference loop (predict_gpu.py:97-159); asynchronous on the current stream."""
        assert images.is_cuda and images.dtype == torch.uint8 and images.is_contiguous()
        n = images.shape[0]
        assert tuple(images.shape[1:]) == (ImageSizeOptions.SEQ_LENGTH, ImageSizeOptions.IMAGE_HEIGHT)
        bases = torch.empty((n, ImageSizeOptions.SEQ_LENGTH), dtype=torch.uint8, device=images.device)
        rles = torch.empty_like(bases)
        acc_b = acc_r = None
        pb = pr = None
        if want_acc:
            acc_b = torch.zeros((n, ImageSizeOptions.SEQ_LENGTH, ImageSizeOptions.TOTAL_BASE_LABELS),
                                dtype=torch.float32, device=images.device)
            acc_r = torch.zeros((n, ImageSizeOptions.SEQ_LENGTH, ImageSizeOptions.TOTAL_RLE_LABELS),
                                dtype=torch.float32, device=images.device)
            pb, pr = acc_b.data_ptr(), acc_r.data_ptr()
        for s in range(0, n, self.max_windows):
            e = min(n, s + self.max_windows)
            _lib.check(self._lib.helen_polish_batch(
                self._handle, images[s:e].data_ptr(), e - s, bases[s:e].data_ptr(),
                rles[s:e].data_ptr(),
                None if pb is None else acc_b[s:e].data_ptr(),
                None if pr is None else acc_r[s:e].data_ptr(), self._stream()))
        if want_acc:
            return bases, rles, acc_b, acc_r
        return bases, rles

    def polish_host(self, images, out=None):
        """images: uint8 numpy / CPU tensor [n,1000,90] -> (bases, rles) numpy u8 [n,1000]; the
        library double-buffers H2D/D2H against compute (helen_polish_host).  `out` = optional
        (bases, rles) C-contiguous uint8 [n,1000] arrays to receive the labels."""
        if isinstance(images, torch.Tensor):
            images = images.numpy()
        images = np.ascontiguousarray(images, dtype=np.uint8)
        n = images.shape[0]
        if out is not None:
            bases, rles = out
            for a in (bases, rles):
                if a.dtype != np.uint8 or a.shape != (n, ImageSizeOptions.SEQ_LENGTH) or not a.flags.c_contiguous:
                    raise ValueError("out arrays must be C-contiguous uint8 [n,1000]")
        else:
            bases = np.empty((n, ImageSizeOptions.SEQ_LENGTH), np.uint8)
            rles = np.empty_like(bases)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.helen_polish_host(
                self._handle, images.ctypes.data, n, bases.ctypes.data, rles.ctypes.data,
                self._stream()))
        return bases, rles

    def submit(self, images, out):
        """The queueing form of polish_host (helen_polish_submit): images uint8 [n,1000,90] (copied before this returns),
        out = (bases, rles) C-contiguous uint8 [n,1000] that receive the labels by the time flush() returns -- keep them
        alive until then.  Device calls go out whenever max_windows windows have gathered."""
        images = np.ascontiguousarray(images.numpy() if isinstance(images, torch.Tensor) else images, dtype=np.uint8)
        n = images.shape[0]
        bases, rles = out
        for a in (bases, rles):
            if a.dtype != np.uint8 or a.shape != (n, ImageSizeOptions.SEQ_LENGTH) or not a.flags.c_contiguous:
                raise ValueError("out arrays must be C-contiguous uint8 [n,1000]")
        with torch.cuda.device(self.device):
            _lib.check(self._lib.helen_polish_submit(self._handle, images.ctypes.data, n, bases.ctypes.data,
                                                     rles.ctypes.data, self._stream()))
        self._queued = getattr(self, "_queued", [])
        self._queued.append((bases, rles))           # (referenced until the flush)

    def flush(self):
        with torch.cuda.device(self.device):
            _lib.check(self._lib.helen_polish_flush(self._handle))
        self._queued = []

    def evaluate(self, images, label_base, label_rle, class_weights, base_confusion, rle_confusion):
        """The per-batch body of the reference's evaluation loop (models/test.py:78-126) for uint8 CUDA
        tensors images [n,1000,90], label_base / label_rle [n,1000].  Returns chunk_stats f32 CUDA
        [n,19,10,3] (sum nll_base, sum w*nll_rle, sum w per window, chunk and group of 10 positions) and
        adds the chunk predictions into the int64 CUDA confusion matrices [5,5] / [11,11]
        ([target][predicted]).  Asynchronous on the current stream."""
        assert images.is_cuda and images.dtype == torch.uint8 and images.is_contiguous()
        n = images.shape[0]
        L = ImageSizeOptions.SEQ_LENGTH
        for lab in (label_base, label_rle):
            assert lab.is_cuda and lab.dtype == torch.uint8 and lab.is_contiguous() and tuple(lab.shape) == (n, L)
        assert base_confusion.dtype == torch.int64 and base_confusion.is_contiguous() and base_confusion.is_cuda
        assert rle_confusion.dtype == torch.int64 and rle_confusion.is_contiguous() and rle_confusion.is_cuda
        cw = np.ascontiguousarray(class_weights, np.float32)
        assert cw.shape == (ImageSizeOptions.TOTAL_RLE_LABELS,)
        stats = torch.empty((n, 19, 10, 3), dtype=torch.float32, device=images.device)
        for s in range(0, n, self.max_windows):
            e = min(n, s + self.max_windows)
            _lib.check(self._lib.helen_evaluate_batch(
                self._handle, images[s:e].data_ptr(), label_base[s:e].data_ptr(), label_rle[s:e].data_ptr(),
                e - s, cw.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), stats[s:e].data_ptr(),
                base_confusion.data_ptr(), rle_confusion.data_ptr(), self._stream()))
        return stats

    def chunk_forward(self, x, hidden):
        """TransducerGRU.forward (TransducerModel.py:60-79): x f32 CUDA [B,T,90], hidden f32 CUDA
        [B,2,128] -> (base [B,T,5], rle [B,T,11], hidden [B,2,128])."""
        assert x.is_cuda and hidden.is_cuda
        x = x.contiguous().float()
        hidden = hidden.contiguous().float()
        B, T, _ = x.shape
        base = torch.empty((B, T, ImageSizeOptions.TOTAL_BASE_LABELS), dtype=torch.float32, device=x.device)
        rle = torch.empty((B, T, ImageSizeOptions.TOTAL_RLE_LABELS), dtype=torch.float32, device=x.device)
        h_out = torch.empty((B, 2, TrainOptions.HIDDEN_SIZE), dtype=torch.float32, device=x.device)
        # rows of a batch never interact: a batch above the engine's capacity goes through in slices, like polish()
        for s in range(0, B, self.max_windows):
            e = min(B, s + self.max_windows)
            _lib.check(self._lib.helen_gru_chunk_forward(
                self._handle, x[s:e].data_ptr(), hidden[s:e].data_ptr(), e - s, T, base[s:e].data_ptr(),
                rle[s:e].data_ptr(), h_out[s:e].data_ptr(), self._stream()))
        return base, rle, h_out

    def reload_overrides(self):
        """Read the environment's A/B switches (HELEN_GRU_PAIR, HELEN_SPLIT, ... : helen_amd/csrc/dispatch.h) again for
        this engine; they are otherwise read once, when it is created."""
        _lib.check(self._lib.helen_reload_overrides(self._handle))

    def inject_failure(self, sub_batch):
        """Test hook: the next polish_host fails right after enqueuing sub-batch `sub_batch` (-1 disarms)."""
        _lib.check(self._lib.helen_debug_inject_failure(self._handle, int(sub_batch)))

    # ---- per-kernel-class timing (HIP events inside the library) ----
    def set_profiling(self, classes):
        mask = 0
        for c in classes:
            mask |= 1 << _lib.KERNEL_CLASSES.index(c)
        _lib.check(self._lib.helen_set_profiling(self._handle, mask))

    def reset_kernel_stats(self):
        _lib.check(self._lib.helen_reset_kernel_stats(self._handle))

    def kernel_stats(self):
        out = {}
        for i, name in enumerate(_lib.KERNEL_CLASSES):
            ms, n = ctypes.c_double(), ctypes.c_longlong()
            _lib.check(self._lib.helen_get_kernel_stats(self._handle, i, ctypes.byref(ms), ctypes.byref(n)))
            out[name] = (ms.value, int(n.value))
        return out
