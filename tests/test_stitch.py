"""stitch: the aligner against the reference's own SSW (oracle/_ref, built from the
reference sources), the anchor logic, the region decode (native == Python restatement), and
end-to-end reconstruction of a known sequence from overlapping regions."""
import ctypes
import os
import random

import numpy as np
import pytest

from helen_amd import hdf5, native_io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SSW = os.path.join(ROOT, "oracle", "_ref", "libssw_ref.so")

pytestmark = pytest.mark.skipif(not (hdf5.available() and native_io.available()),
                                reason="libhdf5 / libhelen_io.so not available")


def _mutate(s, rate, rng):
    out = []
    for c in s:
        x = rng.random()
        if x < rate / 3:
            continue
        if x < 2 * rate / 3:
            out.append(rng.choice("ACGT"))
            out.append(c)
        elif x < rate:
            out.append(rng.choice("ACGTN"))
        else:
            out.append(c)
    return "".join(out)


@pytest.mark.skipif(not os.path.exists(REF_SSW), reason="oracle/_ref/libssw_ref.so not built (make -C oracle ref)")
def test_aligner_equals_reference_ssw_on_random_pairs():
    """Score, begin/end cells, extended CIGAR and mismatch count, cell for cell, with stitch's
    penalties (4, 6, 8, 2; Options.py:4-7).  Score-0 results are compared on the score only: the
    reference library reads out of bounds there and stitch looks at nothing else (Stitch.py:137)."""
    ref = ctypes.CDLL(REF_SSW)
    rng = random.Random(12345)
    checked = 0
    for t in range(2500):
        L = rng.choice([1, 2, 3, 5, 8, 13, 30, 60, 100, 200, 200, 200, 400])
        base = "".join(rng.choice("ACGT") for _ in range(L))
        mode = rng.random()
        if mode < 0.5:
            r, q = base, _mutate(base, rng.choice([0, 0.01, 0.05, 0.15, 0.4]), rng)
            if rng.random() < 0.5 and len(q) > 10:
                k = rng.randrange(0, min(40, len(q) // 2))
                q = q[k:] + "".join(rng.choice("ACGT") for _ in range(k))
        elif mode < 0.7:
            r, q = base, "".join(rng.choice("ACGT") for _ in range(rng.choice([1, 5, 20, 100, 200])))
        elif mode < 0.85:
            r = "".join(rng.choice("AC") * rng.randrange(1, 6) for _ in range(L // 3 + 1))
            q = _mutate(r, 0.1, rng)
        else:
            r, q = _mutate(base, 0.05, rng), base[rng.randrange(0, max(1, L // 2)):]
        if not r or not q:
            continue
        out = (ctypes.c_int * 6)()
        cig = ctypes.create_string_buffer(1 << 16)
        ref.ssw_ref_align(r.encode(), len(r), q.encode(), 4, 6, 8, 2, out, cig, 1 << 16)
        a = native_io.ssw_align(r, q, 4, 6, 8, 2)
        assert a.best_score == out[0], (r, q)
        if out[0] > 0:
            assert (a.reference_begin, a.reference_end, a.query_begin, a.query_end, a.mismatches) == \
                tuple(out[1:6]), (r, q)
            assert a.cigar_string == cig.value.decode(), (r, q)
            checked += 1
    assert checked > 2000


@pytest.mark.skipif(not os.path.exists(REF_SSW), reason="oracle/_ref/libssw_ref.so not built (make -C oracle ref)")
def test_general_aligner_path_equals_reference_ssw_too():
    """Stitch's alignments all take the vectorised 16-bit pass (striped_pass_small); the general pass behind it
    (any length, exact saturation) is forced with HELEN_SSW_GENERAL=1 and put through the same comparison."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", os.path.abspath(__file__), "-k",
                        "test_aligner_equals_reference_ssw_on_random_pairs"],
                       env=dict(os.environ, HELEN_SSW_GENERAL="1"), capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_aligner_known_answers():
    a = native_io.ssw_align("ACGTTGCATGCATGCAAGGCTTAGGACCATTTACGGCATG", "TGCATGCATGCAGGCTTAGGACCTTTTACGGC", 4, 6, 8, 2)
    # produced by the reference library (oracle/_ref) for this pair
    assert (a.best_score, a.reference_begin, a.reference_end, a.query_begin, a.query_end) == (110, 4, 36, 0, 31)
    assert a.cigar_string == "11=1D12=1X8=" and a.mismatches == 2
    z = native_io.ssw_align("AAAA", "", 4, 6, 8, 2)
    assert z.best_score == 0


def test_confident_positions():
    from helen_amd.stitch import get_confident_positions

    class A(object):
        pass
    a = A()
    a.cigar_string, a.reference_begin = "3S5=1X2=4D9=2I7=", 10
    # groups: S3 M8 D4 M9 ...: the first M run >= 8 is the merged 5=1X2= at ref 10, read 3
    assert get_confident_positions(a) == (10, 3)
    a.cigar_string, a.reference_begin = "2S4=1I3=2D5=1X6=", 7
    # M4 I1 M3 D2 M12 -> anchor at ref 7+4+3+2 = 16, read 2+4+1+3 = 10
    assert get_confident_positions(a) == (16, 10)
    a.cigar_string, a.reference_begin = "3=1I3=1D2=", 0
    assert get_confident_positions(a) == (-1, -1)


def _write_predictions(path, contig, regions):
    """regions: list of (start, end, [(chunk_id, positions int64 [n,3], bases, rles)])."""
    from helen_amd.data_store import DataStore
    with DataStore(path, "w") as s:
        for start, end, chunks in regions:
            for chunk_id, pos, bases, rles in chunks:
                n = len(bases)
                P = np.full((1000, 3), -1, np.int64)
                B = np.zeros(1000, np.uint8)
                R = np.zeros(1000, np.uint8)
                P[:n], B[:n], R[:n] = pos, bases, rles
                s.write_prediction(contig, start, end, chunk_id, P, B, R)


def test_region_decode_native_equals_python(tmp_path):
    """First writer wins over chunk ids visited in STRING order ('0','1','10','2',...), numeric key
    order, rle repetition, base 0 = nothing, and the uint32-wrapped padding rows that the
    reference's `< 0` test lets through (they share one key that sorts last)."""
    from helen_amd.stitch import _region_sequence_py
    rng = np.random.default_rng(5)
    chunks = []
    for cid in range(12):                                # ids 0..11 -> '10','11' sort before '2'
        n = 40
        pos = np.stack([rng.integers(0, 60, n), rng.integers(0, 3, n), rng.integers(0, 2, n)], 1)
        chunks.append((cid, pos, rng.integers(0, 5, n), rng.integers(0, 11, n)))
    path = str(tmp_path / "p_0.hdf")
    _write_predictions(path, "ctg", [(0, 1000, chunks)])
    a = native_io.region_sequence(path, "ctg", "ctg-0-1000")
    b = _region_sequence_py(path, "ctg", "ctg-0-1000")
    assert a == b and len(a) > 50
    # the padding rows contribute the first padded row's label once, at the very end
    first = chunks[0]
    tail_base, tail_rle = 0, 0          # padded labels written as zeros by _write_predictions
    assert a.endswith("ACGT"[tail_base - 1] * tail_rle if tail_base else "")


@pytest.mark.parametrize("threads", [1, 3])
def test_stitch_reconstructs_a_known_sequence(tmp_path, threads):
    """Overlapping regions (stride 800, length 1000, like MarginPolish's) whose labels are the run-
    length encoding of a known sequence must stitch back to exactly that sequence; regions are
    spread over two prediction files and images are split into two chunk ids."""
    from helen_amd.stitch import perform_stitch
    rng = random.Random(7)
    runs = []
    prev = None
    while len(runs) < 5000:
        b = rng.choice("ACGT")
        if b == prev:
            continue
        runs.append((b, 1 if rng.random() < 0.7 else rng.randrange(2, 6)))
        prev = b
    truth = "".join(b * r for b, r in runs)
    code = {"A": 1, "C": 2, "G": 3, "T": 4}
    regions = []
    for k in range((len(runs) - 200) // 800):
        lo, hi = 800 * k, min(len(runs), 800 * k + 1000)
        pos = np.stack([np.arange(lo, hi), np.zeros(hi - lo, np.int64), np.zeros(hi - lo, np.int64)], 1)
        bases = np.array([code[b] for b, _ in runs[lo:hi]])
        rles = np.array([r for _, r in runs[lo:hi]])
        half = (hi - lo) // 2
        regions.append((lo, hi, [(0, pos[:half], bases[:half], rles[:half]),
                                 (1, pos[half:], bases[half:], rles[half:])]))
    d = tmp_path / "pred"
    d.mkdir()
    _write_predictions(str(d / "p_0.hdf"), "chrS", regions[0::2])
    _write_predictions(str(d / "p_1.hdf"), "chrS", regions[1::2])
    out = perform_stitch(str(d), str(tmp_path / "out"), "asm", threads)
    lines = open(out).read().split("\n")
    assert lines[0] == ">chrS" and lines[2] == ""
    covered = "".join(b * r for b, r in runs[:regions[-1][1]])
    assert lines[1] == covered and covered == truth[:len(covered)]


def test_stitch_gap_and_no_alignment_paths():
    from helen_amd.stitch import alignment_stitch
    rng = random.Random(3)
    s1 = "".join(rng.choice("ACGT") for _ in range(300))
    s2 = "".join(rng.choice("ACGT") for _ in range(300))
    # no coordinate overlap -> 10 N filler (Stitch.py:180-188)
    c, st, en, seq = alignment_stitch([("c", 0, 300, s1), ("c", 400, 700, s2)])
    assert (st, en) == (0, 700) and seq == s1 + "N" * 10 + s2
    # coordinate overlap but unrelated sequence: either no anchor (filler + whole next chunk) or an
    # anchor by chance; both keep every base of s1 up to the overlap
    c, st, en, seq = alignment_stitch([("c", 0, 300, s1), ("c", 250, 550, s2)])
    assert seq.startswith(s1[:250]) and en == 550
    # a short chunk (<= 10 bases) without an anchor is dropped (Stitch.py:163)
    c, st, en, seq = alignment_stitch([("c", 0, 300, s1), ("c", 400, 405, "ACGTA")])
    assert seq == s1 and en == 300


READERS_CHILD = r'''
import json, sys
sys.path.insert(0, %(root)r)
from helen_amd import native_io
out = {}
for path in %(paths)r:
    for contig in ("ctgA", "ctgB", "nope"):
        listed = native_io.list_regions(path, contig)
        out[path + "|" + contig] = None if listed is None else [
            [name, st, en, native_io.region_sequence(path, contig, name)] for name, st, en in listed]
print(json.dumps(out))
'''


@pytest.mark.parametrize("writer", [None, "libhdf5"])
def test_prediction_readers_agree(tmp_path, monkeypatch, writer):
    """Region listing and region decoding walk prediction files through the direct scanner, with libhdf5 behind it:
    the scanner alone (HELEN_IO_READER=direct), libhdf5 alone and the default must give the same regions, bounds
    and sequences, for files of either writer -- and the same as the Python statement of the decode."""
    import json
    import subprocess
    import sys
    from helen_amd.stitch import _region_sequence_py
    if writer:
        monkeypatch.setenv("HELEN_IO_WRITER", writer)
    else:
        monkeypatch.delenv("HELEN_IO_WRITER", raising=False)
    rng = np.random.default_rng(9)
    paths = []
    for k in range(2):
        path = str(tmp_path / ("p_%d.hdf" % k))
        for contig in (("ctgA", "ctgB") if k == 0 else ("ctgB",)):
            regions = []
            for r in range(23):                              # > 16 members: more than two symbol table nodes
                chunks = []
                for cid in range(1 + r % 3):
                    n = 30 + r
                    pos = np.stack([1000 * r + np.sort(rng.integers(0, 50, n)), rng.integers(0, 2, n), np.zeros(n)], 1)
                    chunks.append((cid, pos.astype(np.int64), rng.integers(0, 5, n), rng.integers(0, 4, n)))
                regions.append((1000 * r, 1000 * r + 1000 + k, chunks))
            # (one DataStore per file: append both contigs of file 0 through the same store)
            if contig == "ctgA" or k == 1:
                from helen_amd.data_store import DataStore
                store = DataStore(path, "w")
            for start, end, chunks in regions:
                for chunk_id, pos, bases, rles in chunks:
                    n = len(bases)
                    P = np.full((1000, 3), -1, np.int64)
                    B = np.zeros(1000, np.uint8)
                    R = np.zeros(1000, np.uint8)
                    P[:n], B[:n], R[:n] = pos, bases, rles
                    store.write_prediction(contig, start, end, chunk_id, P, B, R)
        store.close()
        paths.append(path)
    got = {}
    for reader in ("", "libhdf5", "direct"):
        env = dict(os.environ)
        env.pop("HELEN_IO_READER", None)
        if reader:
            env["HELEN_IO_READER"] = reader
        r = subprocess.run([sys.executable, "-c", READERS_CHILD % {"root": ROOT, "paths": paths}], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[reader] = json.loads(r.stdout.strip().splitlines()[-1])
    assert got[""] == got["libhdf5"] == got["direct"]
    ref = got[""]
    assert ref[paths[0] + "|nope"] is None and ref[paths[1] + "|ctgA"] is None
    assert len(ref[paths[0] + "|ctgA"]) == 23 and len(ref[paths[1] + "|ctgB"]) == 23
    names = [e[0] for e in ref[paths[0] + "|ctgB"]]
    assert names == sorted(names)
    for key, entries in ref.items():
        if entries is None:
            continue
        path, contig = key.split("|")
        for name, st, en, seq in entries[::5]:
            assert name == "%s-%d-%d" % (contig, st, en)
            assert seq == _region_sequence_py(path, contig, name)


def test_signed_positions_of_another_writer_are_skipped_like_the_reference_does(tmp_path):
    """Stitch.py:226 skips rows with a negative pos or indx.  The reference's own writer stores uint32 (its -1 padding
    wraps and is NOT skipped, tested above); a file whose positions are a signed type keeps its negatives, and every
    reader here must then drop those rows like the reference's loop does."""
    import subprocess
    import sys
    from helen_amd.stitch import _region_sequence_py
    path = str(tmp_path / "other.hdf")
    pos = np.array([[5, 0, 0], [-1, -1, -1], [6, 0, 0], [7, -1, 0], [-3, 0, 0], [8, 0, 0]], np.int64)
    bases = np.array([1, 2, 3, 4, 1, 2], np.uint8)
    rles = np.array([1, 3, 2, 2, 5, 1], np.uint8)
    with hdf5.File(path, "w") as f:
        f.write("predictions/c/c-0-9/contig_start", np.int64(0))
        f.write("predictions/c/c-0-9/contig_end", np.int64(9))
        f.write("predictions/c/c-0-9/0/position", pos, np.int64)
        f.write("predictions/c/c-0-9/0/bases", bases, np.uint8)
        f.write("predictions/c/c-0-9/0/rles", rles, np.uint8)
    want = "A" + "GG" + "C"                      # rows 0, 2, 5
    assert _region_sequence_py(path, "c", "c-0-9") == want
    child = ("import sys; sys.path.insert(0, %r); from helen_amd import native_io; "
             "print(native_io.region_sequence(%r, 'c', 'c-0-9'), native_io.list_regions(%r, 'c'))" % (ROOT, path, path))
    for reader in ("", "libhdf5", "direct"):
        env = dict(os.environ)
        env.pop("HELEN_IO_READER", None)
        if reader:
            env["HELEN_IO_READER"] = reader
        r = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        assert r.stdout.strip() == want + " [('c-0-9', 0, 9)]", (reader, r.stdout)


@pytest.mark.skipif(not os.path.exists(REF_SSW), reason="oracle/_ref/libssw_ref.so not built (make -C oracle ref)")
@pytest.mark.parametrize("penalties", [(2, 2, 3, 1), (1, 4, 6, 1), (5, 4, 10, 3), (4, 6, 30, 7)])
def test_aligner_equals_reference_ssw_with_other_penalties(penalties):
    """Stitch only ever passes (4, 6, 8, 2); the entry point takes any penalties, and so does the reference library:
    the library's defaults, a mismatch-heavy set, a match-heavy one and large gap costs."""
    ref = ctypes.CDLL(REF_SSW)
    rng = random.Random(hash(penalties) & 0xffff)
    checked = 0
    for t in range(400):
        L = rng.choice([5, 30, 100, 200, 300])
        base = "".join(rng.choice("ACGT") for _ in range(L))
        r = base
        q = _mutate(base, rng.choice([0, 0.02, 0.1, 0.3]), rng)
        if rng.random() < 0.5 and len(q) > 10:
            k = rng.randrange(0, min(40, len(q) // 2))
            q = q[k:]
        if not q:
            continue
        out = (ctypes.c_int * 6)()
        cig = ctypes.create_string_buffer(1 << 16)
        ref.ssw_ref_align(r.encode(), len(r), q.encode(), *penalties, out, cig, 1 << 16)
        a = native_io.ssw_align(r, q, *penalties)
        assert a.best_score == out[0], (penalties, r, q)
        if out[0] > 0:
            assert (a.reference_begin, a.reference_end, a.query_begin, a.query_end, a.mismatches) == tuple(out[1:6]), \
                (penalties, r, q)
            assert a.cigar_string == cig.value.decode(), (penalties, r, q)
            checked += 1
    assert checked > 300


def test_a_failed_run_fails_the_stitch_and_names_the_incomplete_fasta(tmp_path, monkeypatch):
    """A run whose worker raises (here: a region whose chunk group lacks its datasets): the reference prints the exception
    and returns a FASTA stitched from what survived (StitchInterface.py:40-106); this package writes that FASTA and then
    RAISES, naming it -- unless $HELEN_STITCH_KEEP_GOING=1 asks for the reference's behaviour.  The failure list belongs to
    the call (two calls do not share it)."""
    import helen_amd.stitch as S
    code = {"A": 1, "C": 2, "G": 3, "T": 4}
    seq = "ACGT" * 600
    regions = []
    for k in range(2):
        lo, hi = 800 * k, 800 * k + 1000
        pos = np.stack([np.arange(lo, hi), np.zeros(hi - lo, np.int64), np.zeros(hi - lo, np.int64)], 1)
        regions.append((lo, hi, [(0, pos, np.array([code[c] for c in seq[lo:hi]]), np.ones(hi - lo, np.int64))]))
    good, bad = tmp_path / "good", tmp_path / "bad"
    good.mkdir()
    bad.mkdir()
    _write_predictions(str(good / "p_0.hdf"), "chrS", regions)
    with hdf5.File(str(bad / "p_0.hdf"), "w") as f:          # a region with its bounds and an EMPTY chunk group
        f.write("predictions/chrB/chrB-0-1000/contig_start", 0)
        f.write("predictions/chrB/chrB-0-1000/contig_end", 1000)
        f.write("predictions/chrB/chrB-0-1000/0/position", np.zeros((1000, 3), np.uint32))
    with pytest.raises(RuntimeError, match=r"1 stitch run\(s\) failed, .*asm\.fa is INCOMPLETE"):
        S.perform_stitch(str(bad), str(tmp_path / "out_bad"), "asm", 2)
    assert len(S.FAILED_RUNS) == 1 and os.path.exists(str(tmp_path / "out_bad" / "asm.fa"))
    out = S.perform_stitch(str(good), str(tmp_path / "out_good"), "asm", 2)       # the next call starts clean
    assert S.FAILED_RUNS == [] and open(out).read().startswith(">chrS\n" + seq[:1800])
    monkeypatch.setenv("HELEN_STITCH_KEEP_GOING", "1")
    assert S.perform_stitch(str(bad), str(tmp_path / "out_keep"), "asm", 2).endswith("asm.fa")
    assert len(S.FAILED_RUNS) == 1


def _ref_align(ref, r, q, pen=(4, 6, 8, 2), _buf={}):
    out = _buf.setdefault("out", (ctypes.c_int * 6)())
    cig = _buf.setdefault("cig", ctypes.create_string_buffer(1 << 16))
    ref.ssw_ref_align(r, len(r), q, pen[0], pen[1], pen[2], pen[3], out, cig, 1 << 16)
    return (out[0],) + ((tuple(out[1:6]), cig.value) if out[0] > 0 else ())


def _own_align(lib, r, q, pen=(4, 6, 8, 2), _buf={}):
    out = _buf.setdefault("out", (ctypes.c_int * 6)())
    cig = _buf.setdefault("cig", ctypes.create_string_buffer(1 << 16))
    rc = lib.helen_ssw_align(r, len(r), q, len(q), pen[0], pen[1], pen[2], pen[3], out, cig, 1 << 16)
    assert rc == 0
    return (out[0],) + ((tuple(out[1:6]), cig.value) if out[0] > 0 else ())


def _shortcut_cases(rng):
    """Pairs shaped like stitch's joins and pairs built to break the shortcut's argument: ties between several exact runs,
    tandem repeats that offer a LONGER shifted diagonal, homopolymers, periodic strings, runs at either end of either
    string, one string inside the other, tiny alphabets at tiny lengths (dense in ties)."""
    def rnd(n, alphabet="ACGT"):
        return "".join(rng.choice(alphabet) for _ in range(n))
    kind = rng.random()
    if kind < 0.40:                         # a join: the left string's head is the right string's tail part
        ov = rng.choice([20, 60, 110, 220, 220, 220, 330])
        shared = rng.randrange(max(1, ov // 4), ov + 1)
        mid = rnd(shared)
        left = mid + rnd(ov - shared)
        right = rnd(ov - shared) + mid
        if rng.random() < 0.15:             # one substitution somewhere: the shortcut must usually step aside
            k = rng.randrange(len(right))
            right = right[:k] + rng.choice("ACGT") + right[k + 1:]
        return left, right
    if kind < 0.55:                         # tiny alphabet, tiny strings: every kind of tie
        return rnd(rng.randrange(1, 14), rng.choice(["A", "AC", "ACG", "ACGT"])), rnd(rng.randrange(1, 14), rng.choice(["A", "AC", "ACGT"]))
    if kind < 0.70:                         # tandem repeats and homopolymers under an exact overlap
        unit = rnd(rng.randrange(1, 7))
        rep = unit * rng.randrange(3, 60)
        a, b = rnd(rng.randrange(0, 40)), rnd(rng.randrange(0, 40))
        left = (rep + a)[:rng.randrange(8, 260)]
        right = (b + rep)[-rng.randrange(8, 260):]
        return left, right
    if kind < 0.80:                         # one inside the other, and equal strings
        s = rnd(rng.randrange(1, 300), rng.choice(["ACGT", "AC"]))
        i = rng.randrange(0, len(s))
        j = rng.randrange(i + 1, len(s) + 1)
        return (s, s[i:j]) if rng.random() < 0.5 else (s[i:j], s)
    if kind < 0.90:                         # two copies of the shared run in one of the strings
        mid = rnd(rng.randrange(4, 80))
        gap = rnd(rng.randrange(0, 30))
        return (mid + gap + mid, rnd(rng.randrange(0, 20)) + mid) if rng.random() < 0.5 else (mid + rnd(rng.randrange(0, 20)), mid + gap + mid)
    # low-complexity overlap with the shared part in the middle of both
    mid = rnd(rng.randrange(10, 150), rng.choice(["AC", "ACG", "ACGT"]))
    return rnd(rng.randrange(0, 50)) + mid + rnd(rng.randrange(0, 50)), rnd(rng.randrange(0, 50)) + mid + rnd(rng.randrange(0, 50))


@pytest.mark.skipif(not os.path.exists(REF_SSW), reason="oracle/_ref/libssw_ref.so not built (make -C oracle ref)")
def test_exact_overlap_shortcut_equals_the_reference_library():
    """helen_ssw_align answers a pair whose longest common SUBSEQUENCE is a common SUBSTRING without running the three
    passes (helen_amd/csrc/ssw.cpp: exact_overlap), and one whose forward pass ends on an exact run of score / match bases
    without the other two.  Every such answer -- score, begin / end cells, CIGAR, mismatch count
    -- must be the reference library's own: 120,000 pairs on which a shortcut fires ($HELEN_TEST_SHORTCUT_PAIRS
    sets another number; the round's record run over 200,000: profiles/r06_ssw_shortcut_200k.txt) (stitch-shaped joins and pairs built
    against its argument), each also run with the shortcut off."""
    ref = ctypes.CDLL(REF_SSW)
    lib = native_io.load()
    rng = random.Random(20260930)
    fired = tried = 0
    after_forward0 = lib.helen_ssw_fast_path_after_forward()
    before = native_io.ssw_fast_path(True)
    try:
        target = int(os.environ.get("HELEN_TEST_SHORTCUT_PAIRS", "120000"))
        while fired < target and tried < 8 * target:
            r, q = _shortcut_cases(rng)
            if not r or not q:
                continue
            tried += 1
            rb, qb = r.encode(), q.encode()
            h0 = native_io.ssw_fast_path_counts()[0]
            got = _own_align(lib, rb, qb)
            if native_io.ssw_fast_path_counts()[0] == h0:
                continue                     # the three passes answered: covered by the tests above
            fired += 1
            want = _ref_align(ref, rb, qb)
            assert got == want, (r, q, got, want)
            if fired % 16 == 0:              # ... and this library's own three passes say the same
                native_io.ssw_fast_path(False)
                slow = _own_align(lib, rb, qb)
                native_io.ssw_fast_path(True)
                assert slow == got, (r, q, slow, got)
    finally:
        native_io.ssw_fast_path(before)
    assert fired >= target, (fired, tried)
    second = lib.helen_ssw_fast_path_after_forward() - after_forward0
    print("exact-overlap shortcuts: %d of %d pairs answered by them (%d only after the forward pass), all equal to the reference "
          "library" % (fired, tried, second))
    assert second > 5000 and fired - second > 30000          # both shortcuts are exercised


@pytest.mark.skipif(not os.path.exists(REF_SSW), reason="oracle/_ref/libssw_ref.so not built (make -C oracle ref)")
@pytest.mark.parametrize("penalties", [(2, 2, 3, 1), (1, 4, 6, 1), (5, 4, 10, 3), (1, 1, 1, 0)])
def test_exact_overlap_shortcut_with_other_penalties(penalties):
    """The argument needs match > 0 and mismatch, gap_open > 0 only: other penalty sets, gap_extend = 0 included."""
    ref = ctypes.CDLL(REF_SSW)
    lib = native_io.load()
    rng = random.Random(sum(penalties) * 7919)
    fired = 0
    for _ in range(30000):
        r, q = _shortcut_cases(rng)
        if not r or not q:
            continue
        h0 = native_io.ssw_fast_path_counts()[0]
        got = _own_align(lib, r.encode(), q.encode(), penalties)
        if native_io.ssw_fast_path_counts()[0] != h0:
            fired += 1
            assert got == _ref_align(ref, r.encode(), q.encode(), penalties), (penalties, r, q)
    assert fired > 5000
