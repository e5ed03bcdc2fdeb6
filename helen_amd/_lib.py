"""ctypes binding of libhelen_hip.so (C ABI in include/helen_hip.h).

The HIP library is the product: there is no CPU fallback.  If the shared object is missing or
fails to load, importing the compute entry points raises -- loudly -- rather than degrading.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HELEN_HIP_LIB: developer override to A/B-test kernel variants built side by side.
LIB_PATH = os.environ.get("HELEN_HIP_LIB") or os.path.join(_HERE, "csrc", "libhelen_hip.so")

HELEN_ABI_VERSION = 4
HELEN_OK = 0
HELEN_PRECISION_FP32 = 0
HELEN_PRECISION_BF16 = 1
HELEN_PRECISION_FP32X3 = 2

KERNEL_CLASSES = ("pack", "gemm_enc", "gru_enc", "gemm_dec", "gru_dec", "heads")

# every symbol include/helen_hip.h declares
EXPORTS = (
    "helen_abi_version", "helen_last_error", "helen_model_create", "helen_model_destroy",
    "helen_model_device_bytes", "helen_polish_batch", "helen_polish_host", "helen_polish_submit", "helen_polish_flush",
    "helen_gru_chunk_forward", "helen_evaluate_batch", "helen_debug_inject_failure", "helen_set_profiling",
    "helen_reset_kernel_stats",
    "helen_get_kernel_stats", "helen_reload_overrides", "helen_describe_dispatch", "helen_plan_call",
    "helen_device_count", "helen_host_alloc", "helen_host_free", "helen_polish_slot_submit", "helen_polish_slot_wait",
)
RECURRENCE_KERNELS = ("gru_kernel", "gru_single8_kernel", "gru_half8_kernel", "gru_quarter4_kernel", "gru_pair_kernel")
DECODER_PROJECTIONS = ("gemm_gi_kernel<16>", "gemm_dec_ws_kernel", "gemm_dec_wsp_kernel")
ENCODER_PROJECTIONS = ("gemm_enc_x3_kernel",)        # polish entry points: exact bf16 products, fp32 accumulation

_f32p = ctypes.POINTER(ctypes.c_float)


class HelenWeightsC(ctypes.Structure):
    """`HelenWeights` of include/helen_hip.h."""
    _fields_ = [
        ("features", ctypes.c_int32), ("hidden", ctypes.c_int32),
        ("n_base", ctypes.c_int32), ("n_rle", ctypes.c_int32),
        ("enc_w_ih", _f32p * 2), ("enc_w_hh", _f32p * 2),
        ("enc_b_ih", _f32p * 2), ("enc_b_hh", _f32p * 2),
        ("dec_w_ih", _f32p * 2), ("dec_w_hh", _f32p * 2),
        ("dec_b_ih", _f32p * 2), ("dec_b_hh", _f32p * 2),
        ("base_w", _f32p), ("base_b", _f32p), ("rle_w", _f32p), ("rle_b", _f32p),
    ]


class HelenError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libhelen_hip error %d: %s" % (code, message))
        self.code = code


_lib = None


def _try_build(target):
    """The .so files are build products kept out of git; if one is missing (fresh clone) and the
    toolchain is here, build it in-tree once.  Failure is silent here -- the caller raises."""
    import shutil
    import subprocess
    if shutil.which("make") and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        try:
            subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), target], check=False,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        except Exception:
            pass


def load(with_torch=True):
    """Load libhelen_hip.so once; raise if it is absent (build with `python __graft_entry__.py`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) and not os.environ.get("HELEN_HIP_LIB"):
        _try_build("libhelen_hip.so")
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "helen_amd: %s not found. The HIP library is required (there is no CPU fallback); "
            "build it with `make -C helen_amd/csrc` or `python __graft_entry__.py`." % LIB_PATH)
    # PyTorch-ROCm carries its own HIP runtime; import it first so that this process has ONE
    # libamdhip64 (device buffers and streams are torch's and are handed to the library by pointer).  A process that
    # never touches torch -- `helen polish` on its native slot pipeline, helen_amd.predict -- asks for with_torch=False
    # and runs on the system's runtime; it must then not import torch afterwards (helen_amd.predict decides up front).
    import sys
    if with_torch or "torch" in sys.modules:
        import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.helen_abi_version.restype = ci
    lib.helen_abi_version.argtypes = []
    lib.helen_last_error.restype = ctypes.c_char_p
    lib.helen_last_error.argtypes = []
    lib.helen_model_create.restype = ci
    lib.helen_model_create.argtypes = [ctypes.POINTER(HelenWeightsC), ci, ci, ci,
                                       ctypes.POINTER(vp)]
    lib.helen_model_destroy.restype = ci
    lib.helen_model_destroy.argtypes = [vp]
    lib.helen_model_device_bytes.restype = ci
    lib.helen_model_device_bytes.argtypes = [vp, ctypes.POINTER(ctypes.c_size_t)]
    lib.helen_polish_batch.restype = ci
    lib.helen_polish_batch.argtypes = [vp, vp, ci, vp, vp, vp, vp, vp]
    lib.helen_polish_host.restype = ci
    lib.helen_polish_host.argtypes = [vp, vp, ci, vp, vp, vp]
    lib.helen_polish_submit.restype = ci
    lib.helen_polish_submit.argtypes = [vp, vp, ci, vp, vp, vp]
    lib.helen_polish_flush.restype = ci
    lib.helen_polish_flush.argtypes = [vp]
    lib.helen_gru_chunk_forward.restype = ci
    lib.helen_gru_chunk_forward.argtypes = [vp, vp, vp, ci, ci, vp, vp, vp, vp]
    lib.helen_evaluate_batch.restype = ci
    lib.helen_evaluate_batch.argtypes = [vp, vp, vp, vp, ci, _f32p, vp, vp, vp, vp]
    lib.helen_debug_inject_failure.restype = ci
    lib.helen_debug_inject_failure.argtypes = [vp, ci]
    lib.helen_set_profiling.restype = ci
    lib.helen_set_profiling.argtypes = [vp, ctypes.c_uint]
    lib.helen_reset_kernel_stats.restype = ci
    lib.helen_reset_kernel_stats.argtypes = [vp]
    lib.helen_get_kernel_stats.restype = ci
    lib.helen_get_kernel_stats.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_double),
                                           ctypes.POINTER(ctypes.c_longlong)]
    lib.helen_reload_overrides.restype = ci
    lib.helen_reload_overrides.argtypes = [vp]
    lib.helen_describe_dispatch.restype = ci
    lib.helen_describe_dispatch.argtypes = [ci, ctypes.c_char_p, ctypes.c_size_t]
    lib.helen_plan_call.restype = ci
    lib.helen_plan_call.argtypes = [ci, ci, ctypes.POINTER(ci)]
    lib.helen_device_count.restype = ci
    lib.helen_device_count.argtypes = [ctypes.POINTER(ci)]
    lib.helen_host_alloc.restype = ci
    lib.helen_host_alloc.argtypes = [ci, ctypes.c_size_t, ctypes.POINTER(vp)]
    lib.helen_host_free.restype = ci
    lib.helen_host_free.argtypes = [vp]
    lib.helen_polish_slot_submit.restype = ci
    lib.helen_polish_slot_submit.argtypes = [vp, vp, ci, vp, vp, vp]
    lib.helen_polish_slot_wait.restype = ci
    lib.helen_polish_slot_wait.argtypes = [vp]
    got = lib.helen_abi_version()
    if got != HELEN_ABI_VERSION:
        raise ImportError("libhelen_hip.so ABI %d != binding ABI %d; rebuild" % (got, HELEN_ABI_VERSION))
    _lib = lib
    return lib


def describe_dispatch(cus):
    """The kernel table of a device with `cus` compute units (helen_amd/csrc/dispatch.h), as text: a dry run."""
    buf = ctypes.create_string_buffer(1 << 16)
    check(load().helen_describe_dispatch(int(cus), buf, len(buf)))
    return buf.value.decode()


def plan_call(cus, tiles):
    """What a call of `tiles` tiles takes on `cus` CUs: dict(split, first_group, recurrence, decoder, decoder_runs, encoder,
    encoder_runs, bf16_two_tiles) with the kernels by name."""
    out = (ctypes.c_int * 8)()
    check(load().helen_plan_call(int(cus), int(tiles), out))
    return {"split": bool(out[0]), "first_group": int(out[1]), "recurrence": RECURRENCE_KERNELS[out[2]],
            "decoder": DECODER_PROJECTIONS[out[3]], "decoder_runs": int(out[4]), "encoder": ENCODER_PROJECTIONS[out[5]],
            "encoder_runs": int(out[6]), "bf16_two_tiles": bool(out[7])}


def check(rc):
    if rc != HELEN_OK:
        raise HelenError(rc, load().helen_last_error().decode("utf-8", "replace"))
