"""stitch against tests/naive_stitch.py -- a second, independent, deliberately naive statement
of Stitch.py:34-255 that takes its alignments from the reference's own SSW build when present: fuzzed joins
(mutated overlaps, overlaps shorter than the anchor run, no overlap, nested and tiny chunks), random CIGARs,
region decoding with uint32-wrapped padding rows and duplicate keys across chunk ids in string order, and whole
prediction directories (several files, contigs, a single-region contig, 1 and 3 workers)."""
import os
import random

import numpy as np
import pytest

import naive_stitch
from helen_amd import hdf5, native_io

pytestmark = pytest.mark.skipif(not (hdf5.available() and native_io.available()),
                                reason="libhdf5 / libhelen_io.so not available")


def _rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def _mutate(rng, s, rate):
    out = []
    for c in s:
        x = rng.random()
        if x < rate / 3:
            continue                                   # deletion
        if x < 2 * rate / 3:
            out.append(rng.choice("ACGT"))             # insertion
            out.append(c)
        elif x < rate:
            out.append(rng.choice("ACGTN"))            # substitution
        else:
            out.append(c)
    return "".join(out)


def _chain(rng):
    """2-5 chunks cut from one truth sequence with every kind of trouble at the seams."""
    truth = _rand_seq(rng, rng.choice([60, 300, 1200]))
    chunks, start = [], 0
    for k in range(rng.randrange(2, 6)):
        length = rng.choice([3, 9, 11, 40, 150, 400])
        end = min(len(truth), start + length)
        seq = truth[start:end]
        kind = rng.random()
        if kind < 0.35:
            seq = _mutate(rng, seq, rng.choice([0.01, 0.05, 0.2, 0.5]))
        elif kind < 0.45:
            seq = _rand_seq(rng, len(seq))             # unrelated: no anchor / no alignment
        elif kind < 0.5:
            seq = ""                                    # an empty region (Stitch.py:141-147)
        chunks.append(("ctg", start, end, seq))
        step = rng.random()
        if step < 0.55:
            start = max(start + 1, end - rng.choice([1, 3, 7, 8, 9, 20, 60, 200]))   # overlap, some below the run of 8
        elif step < 0.7:
            start = end                                 # abutting: no coordinate overlap
        elif step < 0.85:
            start = end + rng.choice([1, 50])           # a gap
        else:
            start = start + rng.choice([1, 5])          # nested inside the previous chunk
        if start >= len(truth):
            break
    rng.shuffle(chunks)                                 # the procedure sorts
    return chunks


def test_fuzzed_joins_equal_the_naive_statement(capfd):
    from helen_amd.stitch import alignment_stitch
    rng = random.Random(20260928)
    seen = {"anchor": 0, "filler": 0}
    for _ in range(1500):
        chunks = _chain(rng)
        got = alignment_stitch(list(chunks))
        want = naive_stitch.join(list(chunks))
        assert got == want, chunks
        seen["filler" if "N" * 10 in got[3] else "anchor"] += 1
    capfd.readouterr()                                  # the procedure warns on stderr: drop it
    assert seen["anchor"] > 300 and seen["filler"] > 300


def test_random_cigars_give_the_same_anchor():
    from helen_amd.stitch import get_confident_positions

    class A(object):
        pass
    rng = random.Random(5)
    hits = 0
    for _ in range(4000):
        parts, last = [], None
        for _ in range(rng.randrange(1, 9)):
            op = rng.choice([o for o in "=XIDSM" if o != last or o in "=X"])
            parts.append("%d%s" % (rng.choice([1, 2, 3, 5, 7, 8, 9, 30]), op))
            last = op
        a = A()
        a.cigar_string = "".join(parts)
        a.reference_begin = rng.randrange(0, 50)
        got = get_confident_positions(a)
        assert got == naive_stitch.anchor(a.reference_begin, a.cigar_string), a.cigar_string
        hits += got != (-1, -1)
    assert 500 < hits < 3500
    a.cigar_string = "5=3N9="
    with pytest.raises(ValueError):
        get_confident_positions(a)


def _write(path, contig, regions):
    """regions: [(start, end, [(chunk_id, positions [n,3] (may hold -1 rows), bases, rles)])] through DataStore,
    which stores positions as uint32 (the -1 padding wraps to 4294967295, DataStore.py:126)."""
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


def test_region_decoding_equals_the_naive_statement(tmp_path):
    """Twelve chunk ids ('10' and '11' are visited before '2'), keys repeated across and inside images with
    different labels, explicit -1 rows in the middle of an image, padded tails with NON-zero labels."""
    from helen_amd.data_store import DataStore
    rng = np.random.default_rng(11)
    path = str(tmp_path / "p_0.hdf")
    with DataStore(path, "w") as s:
        for region in range(6):
            for cid in range(12 if region else 1):
                n = int(rng.integers(1, 1000))
                P = np.full((1000, 3), -1, np.int64)
                P[:n] = np.stack([rng.integers(0, 80, n), rng.integers(0, 3, n), rng.integers(0, 2, n)], 1)
                holes = rng.integers(0, n, 5)
                P[holes] = -1                            # padding rows in the middle
                B = rng.integers(0, 5, 1000).astype(np.uint8)     # labels under the padding rows too
                R = rng.integers(0, 11, 1000).astype(np.uint8)
                s.write_prediction("ctg", 800 * region, 800 * region + 1000, cid, P, B, R)
    with hdf5.File(path, "r") as f:
        for region in range(6):
            name = "ctg-%d-%d" % (800 * region, 800 * region + 1000)
            want = naive_stitch.decode_region(f, "ctg", name)
            assert native_io.region_sequence(path, "ctg", name) == want
            assert len(want) > 20


@pytest.mark.parametrize("threads", [1, 3])
def test_directories_equal_the_naive_statement(tmp_path, threads, capfd):
    from helen_amd.stitch import perform_stitch
    rng = random.Random(99)
    code = {"A": 1, "C": 2, "G": 3, "T": 4}
    d = tmp_path / "pred"
    d.mkdir()
    per_file = {0: {}, 1: {}, 2: {}}
    for contig, n_regions in (("chrA", 9), ("chrB", 1), ("chrC", 4)):
        runs, prev = [], None
        while len(runs) < 800 * n_regions + 400:
            b = rng.choice("ACGT")
            if b != prev:
                runs.append((b, 1 if rng.random() < 0.7 else rng.randrange(2, 7)))
                prev = b
        for k in range(n_regions):
            lo, hi = 800 * k, 800 * k + 1000
            if contig == "chrC" and k == 2:
                lo += 300                                 # a hole: regions 1 and 2 do not overlap
            mine = [(b, r) for b, r in runs[lo:hi]]
            if rng.random() < 0.4:                         # errors in this region's calls
                mine = [(rng.choice("ACGT"), r) if rng.random() < 0.03 else (b, max(0, r + rng.choice([-1, 0, 0, 0, 1])))
                        for b, r in mine]
            pos = np.stack([np.arange(lo, hi), np.zeros(hi - lo, np.int64), np.zeros(hi - lo, np.int64)], 1)
            bases = np.array([code[b] for b, _ in mine])
            rles = np.array([r for _, r in mine])
            cut = rng.randrange(100, 900 - (lo % 800))
            chunks = [(0, pos[:cut], bases[:cut], rles[:cut]), (1, pos[cut - 20:], bases[cut - 20:], rles[cut - 20:])]
            per_file[rng.randrange(3)].setdefault(contig, []).append((lo, hi, chunks))
    for k, contigs in per_file.items():
        if contigs:
            from helen_amd.data_store import DataStore
            path = str(d / ("p_%d.hdf" % k))
            with DataStore(path, "w") as s:
                for contig, regions in contigs.items():
                    for start, end, chunks in regions:
                        for chunk_id, pos, bases, rles in chunks:
                            n = len(bases)
                            P = np.full((1000, 3), -1, np.int64)
                            B = np.zeros(1000, np.uint8)
                            R = np.zeros(1000, np.uint8)
                            P[:n], B[:n], R[:n] = pos, bases, rles
                            s.write_prediction(contig, start, end, chunk_id, P, B, R)
    out = perform_stitch(str(d), str(tmp_path / "fa"), "asm", threads)
    capfd.readouterr()
    lines = open(out).read().split("\n")
    got = {lines[i][1:]: lines[i + 1] for i in range(0, len(lines) - 1, 2)}
    want = naive_stitch.stitch_directory(str(d), threads)
    assert set(got) == set(want) == {"chrA", "chrB", "chrC"}
    for contig in want:
        assert got[contig] == want[contig], contig
    assert "N" * 10 in got["chrC"]                          # the hole was filled, in both


def _reference_fixture():
    import gzip
    import json
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stitch_ref.json.gz")
    with gzip.open(path, "rt") as f:
        return json.load(f)


def test_joins_equal_the_reference_stitch_itself(capfd):
    """tests/golden/stitch_ref.json.gz holds what the REFERENCE's own Stitch.alignment_stitch (Stitch.py:96-190), imported
    from the reference tree and run on the reference's own SSW library, returned for 400 seeded chains of chunks
    (tests/golden/make_golden_stitch.py).  This package's alignment_stitch -- its own aligner included -- and the naive
    statement must return exactly that."""
    from helen_amd.stitch import alignment_stitch
    cases = _reference_fixture()["joins"]
    assert len(cases) == 400
    fillers = 0
    for case in cases:
        chunks = [tuple(c) for c in case["chunks"]]
        want = tuple(case["result"])
        assert alignment_stitch(list(chunks)) == want, chunks
        assert naive_stitch.join(list(chunks)) == want, chunks
        fillers += "N" * 10 in want[3]
    capfd.readouterr()
    assert 100 < fillers < 300


def test_anchors_equal_the_reference_stitch_itself():
    """... and Stitch.get_confident_positions (Stitch.py:33-94) on 1,500 seeded CIGARs."""
    from helen_amd.stitch import get_confident_positions

    class A(object):
        pass
    cases = _reference_fixture()["anchors"]
    assert len(cases) == 1500
    hits = 0
    for case in cases:
        a = A()
        a.cigar_string, a.reference_begin = case["cigar"], case["reference_begin"]
        if case["result"] == "ValueError":
            with pytest.raises(ValueError):
                get_confident_positions(a)
            continue
        assert list(get_confident_positions(a)) == case["result"], case
        assert list(naive_stitch.anchor(a.reference_begin, a.cigar_string)) == case["result"], case
        hits += case["result"][0] != -1
    assert hits > 200


@pytest.mark.parametrize("writer", [None, "libhdf5"])
def test_region_decode_and_contig_stitch_equal_the_reference_stitch_itself(tmp_path, monkeypatch, capfd, writer):
    """14 prediction directories (seeded regions: duplicate keys across chunk ids in string order, insert columns, gap
    labels, padding rows, noisy regions, holes) on which the REFERENCE's own Stitch.small_chunk_stitch and
    create_consensus_sequence and the whole StitchInterface.perform_stitch (1 and 3 workers, one or two contigs) were run
    (tests/golden/make_golden_stitch.py).  Rebuilt here from the fixture's rows, through either writer, this package's
    small_chunk_stitch / create_consensus_sequence / perform_stitch and the naive statement must give the reference's
    sequences and the reference's FASTA file."""
    import importlib.util
    from helen_amd.stitch import create_consensus_sequence, perform_stitch, small_chunk_stitch
    if writer:
        monkeypatch.setenv("HELEN_IO_WRITER", writer)
    else:
        monkeypatch.delenv("HELEN_IO_WRITER", raising=False)
    spec = importlib.util.spec_from_file_location(
        "make_golden_stitch", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden_stitch.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)                         # only its write_case helper is used: nothing of the reference
    cases = _reference_fixture()["directories"]
    assert len(cases) == 14
    for k, case in enumerate(cases):
        d = tmp_path / ("case%d" % k)
        d.mkdir()
        gen.write_case(str(d), case)
        for threads in (1, 3):            # the reference's whole driver, StitchInterface.perform_stitch: the FASTA itself
            out = perform_stitch(str(d), str(d / ("fasta%d" % threads)), "asm", threads)
            assert open(out).read() == case["perform_stitch"][str(threads)], (k, threads)
        keys = sorted((("ctg", str(d / r["file"]), "ctg-%d-%d" % (r["start"], r["end"]), r["start"], r["end"])
                       for r in case["regions"] if r["contig"] == "ctg"), key=lambda e: (e[3], e[4]))
        got = small_chunk_stitch("ctg", keys)
        assert [got[0], got[1], got[2], got[3].decode() if isinstance(got[3], bytes) else got[3]] == \
            case["small_chunk_stitch"], k
        tuples = [(key[1], key[2], key[3], key[4]) for key in keys]
        for threads in (1, 3):
            seq = create_consensus_sequence("ctg", tuples, threads)
            assert bytes(seq).decode() == case["create_consensus_sequence"][str(threads)], (k, threads)
        assert naive_stitch.stitch_directory(str(d), threads=3)["ctg"] == case["create_consensus_sequence"]["3"], k
    capfd.readouterr()
