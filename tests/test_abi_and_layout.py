"""The C-ABI library loads and exports every symbol include/helen_hip.h declares (no compute
without a GPU); the product never touches the oracle; the CPU-only failure mode is loud."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from helen_amd import _lib
    header = open(os.path.join(ROOT, "include", "helen_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(helen_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.helen_abi_version() == _lib.HELEN_ABI_VERSION
    assert lib.helen_last_error() == b""


def test_io_library_exports_exactly_what_its_header_declares():
    """include/helen_io.h <-> libhelen_io.so: every declared entry point is exported, and nothing named helen_* is
    exported that the header does not declare (both sources include the header, so the signatures are the compiler's
    business)."""
    import subprocess

    from helen_amd import native_io
    if not native_io.available():
        pytest.skip("libhelen_io.so not available")
    header = open(os.path.join(ROOT, "include", "helen_io.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(helen_[a-z_0-9]+)\s*\(", header))
    out = subprocess.run(["nm", "-D", "--defined-only", native_io.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\b[TB] (helen_[a-z_0-9]+)$", out, flags=re.M))
    assert declared == exported, declared ^ exported
    lib = native_io.load()
    assert lib.helen_io_abi_version() == 1


def test_abi_rejects_bad_arguments_without_a_gpu():
    import ctypes

    from helen_amd import _lib
    from helen_amd.engine import weights_struct
    from helen_amd.weights import make_weights
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.helen_model_create(None, 0, 16, 0, ctypes.byref(h)) == -1        # HELEN_EINVAL
    assert b"null" in lib.helen_last_error()
    s, keep = weights_struct(make_weights())
    s.features = 10                                                            # not the model's F
    assert lib.helen_model_create(ctypes.byref(s), 0, 16, 0, ctypes.byref(h)) == -1
    assert b"unsupported geometry" in lib.helen_last_error()
    s.features = 90
    assert lib.helen_model_create(ctypes.byref(s), 0, 0, 0, ctypes.byref(h)) == -1
    assert lib.helen_polish_batch(None, None, 1, None, None, None, None, None) == -1


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from helen_amd.engine import HelenEngine
    from helen_amd.weights import make_weights
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HelenEngine(make_weights())


def test_product_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "helen_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                for line in src.splitlines():
                    if re.match(r"\s*(from|import)\s+oracle\b", line) or "libhelen_oracle" in line \
                            or '#include "../../oracle' in line:
                        raise AssertionError("%s references the oracle: %s" % (f, line))


def test_chunk_starts_match_reference_loop():
    from helen_amd.options import chunk_starts
    assert chunk_starts() == list(range(0, 901, 50)) and len(chunk_starts()) == 19
