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


def test_every_environment_switch_is_in_the_readme_table():
    """README.md's table of HELEN_* variables (name, default, supported / probe) and the names the code reads agree:
    every name passed to getenv / flag_of in csrc/ and every HELEN_* read from os.environ in the package, bin/ and
    bench.py has a row, and no row names a variable nothing reads."""
    read = set()
    csrc = os.path.join(ROOT, "helen_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".h", ".hip", ".cpp")):
            read |= set(re.findall(r'(?:getenv|flag_of)\("(HELEN_[A-Z0-9_]+)"\)', open(os.path.join(csrc, f)).read()))
    py = [os.path.join(ROOT, "bench.py")]
    py += [os.path.join(ROOT, "bin", f) for f in os.listdir(os.path.join(ROOT, "bin"))]
    py += [os.path.join(ROOT, "helen_amd", f) for f in os.listdir(os.path.join(ROOT, "helen_amd")) if f.endswith(".py")]
    for f in py:
        src = open(f).read()
        read |= set(re.findall(r'environ(?:\.get\(|\[|\.setdefault\(|\.pop\()\s*["\'](HELEN_[A-Z0-9_]+)', src))
        read |= set(re.findall(r'["\'](HELEN_[A-Z0-9_]+)["\']\s+(?:not\s+)?in\s+os\.environ', src))
    readme = open(os.path.join(ROOT, "README.md")).read()
    table = readme[readme.index("## Environment variables"):]
    rows = set(re.findall(r"HELEN_[A-Z0-9_]+", table))
    assert len(read) > 20, sorted(read)
    assert not (read - rows), "read by the code, no row in README.md: %s" % sorted(read - rows)
    assert not (rows - read), "in README.md's table, read by nothing: %s" % sorted(rows - read)


def test_chunk_starts_match_reference_loop():
    from helen_amd.options import chunk_starts
    assert chunk_starts() == list(range(0, 901, 50)) and len(chunk_starts()) == 19


def test_dispatch_table_dry_run(monkeypatch):
    """helen_amd/csrc/dispatch.h: which kernels a call takes is ONE table derived from the device's CU count -- run dry
    here (no device) for the MI355X's 256 CUs and for 304- and 128-CU parts.  The 256-CU rows are DESIGN.md 6's table;
    the other two must be the same rules scaled, with no tile count left without a plan; an environment switch changes
    exactly its own column."""
    from helen_amd import _lib
    for k in ("HELEN_GRU_PAIR", "HELEN_GRU_SINGLE8", "HELEN_GRU_HALF8", "HELEN_GRU_QUARTER4", "HELEN_DEC_WS", "HELEN_DEC_WSP",
              "HELEN_SPLIT", "HELEN_SPLIT_AT", "HELEN_BF16_PAIR"):
        monkeypatch.delenv(k, raising=False)
    want = {1: "gru_quarter4_kernel", 32: "gru_quarter4_kernel", 33: "gru_half8_kernel", 64: "gru_half8_kernel",
            86: "gru_single8_kernel", 128: "gru_single8_kernel", 240: "gru_pair_kernel", 256: "gru_pair_kernel"}
    for tiles, rec in want.items():
        p = _lib.plan_call(256, tiles)
        assert p["recurrence"] == rec and not p["split"], (tiles, p)
    assert _lib.plan_call(256, 256)["decoder"] == "gemm_dec_ws_kernel" and _lib.plan_call(256, 256)["encoder_runs"] == 1
    assert _lib.plan_call(256, 16)["decoder"] == "gemm_dec_wsp_kernel" and _lib.plan_call(256, 16)["decoder_runs"] == 7
    assert _lib.plan_call(256, 64)["decoder_runs"] == 2 and _lib.plan_call(256, 100)["decoder"] == "gemm_gi_kernel<16>"
    for tiles, first in ((65, 64), (85, 64), (129, 65), (144, 72), (160, 128), (192, 128), (239, 128)):
        p = _lib.plan_call(256, tiles)
        assert p["split"] and p["first_group"] == first, (tiles, p)
    assert not _lib.plan_call(256, 86)["split"] and not _lib.plan_call(256, 240)["split"]
    for cus in (128, 304):
        text = _lib.describe_dispatch(cus)
        assert text.splitlines()[0].startswith("dispatch for %d CUs" % cus)
        # the same rules in CUs: an eighth / a quarter / half of the CUs in tiles, the split ranges, whole rounds of pairs
        assert _lib.plan_call(cus, cus // 8)["recurrence"] == "gru_quarter4_kernel"
        assert _lib.plan_call(cus, cus // 8 + 1)["recurrence"] == "gru_half8_kernel"
        assert _lib.plan_call(cus, cus // 4)["recurrence"] == "gru_half8_kernel"
        assert _lib.plan_call(cus, cus // 4 + 1)["split"] and _lib.plan_call(cus, cus // 4 + 1)["first_group"] == cus // 4
        assert _lib.plan_call(cus, cus // 2)["recurrence"] == "gru_single8_kernel" and not _lib.plan_call(cus, cus // 2)["split"]
        assert _lib.plan_call(cus, cus // 2 + 1)["split"]
        assert _lib.plan_call(cus, cus)["recurrence"] == "gru_pair_kernel" and not _lib.plan_call(cus, cus)["split"]
        assert _lib.plan_call(cus, cus)["decoder"] == "gemm_dec_ws_kernel" and _lib.plan_call(cus, cus)["encoder"] == "gemm_enc_x3_kernel"
        for tiles in range(1, 2 * cus + 1):                  # every size has a plan whose groups are smaller than the call
            p = _lib.plan_call(cus, tiles)
            assert not p["split"] or 0 < p["first_group"] < tiles, (cus, tiles, p)
    # switches: read from the environment at model creation (here: at the dry run)
    monkeypatch.setenv("HELEN_GRU_PAIR", "0")
    monkeypatch.setenv("HELEN_SPLIT", "0")
    p = _lib.plan_call(256, 256)
    assert p["recurrence"] == "gru_kernel" and p["decoder"] == "gemm_dec_ws_kernel"
    assert not _lib.plan_call(256, 192)["split"]
    monkeypatch.setenv("HELEN_SPLIT", "1")
    monkeypatch.setenv("HELEN_SPLIT_AT", "5")
    assert _lib.plan_call(256, 20) == dict(_lib.plan_call(256, 20), split=True, first_group=5)
    # the product reads no switch per call: api.hip has exactly one getenv (the debug-hook gate), dispatch.h the rest
    src = open(os.path.join(ROOT, "helen_amd", "csrc", "api.hip")).read()
    assert src.count("getenv(") == 1 and "use_pair_recurrence" not in src and "use_split" not in src

