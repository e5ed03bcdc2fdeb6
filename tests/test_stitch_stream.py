"""The stitch stage that runs behind the inference (helen_amd/stitch_stream.py) against the two-phase stitch it replaces
in `polish`: whatever arrives in whatever batches, the FASTA must be the one `perform_stitch` (pinned to the reference's
Stitch.py by tests/test_stitch.py) writes from the finished prediction files."""
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from helen_amd import native_io  # noqa: E402

pytestmark = pytest.mark.skipif(not native_io.available(), reason="libhelen_io.so is not built")


def _windows(rng, n_contigs=3, noisy_every=7):
    """Random prediction windows in writing order: (contig, start, end, chunk id, positions [1000,3], bases, rles).
    Regions overlap their neighbours by 150-250 truth bases and carry the truth there (so joins align), except the noisy
    ones; chunk ids run past 9 (string order), some images repeat keys of the image before with other labels (first writer
    wins), some are repeated whole (the writer skips them), some regions are empty or shorter than ten bases, one contig
    has a hole, and a few regions' images are NOT back to back."""
    out = []
    for c in range(n_contigs):
        contig = "ctg%d.%s" % (c, "x" * (3 * c))
        length = rng.randrange(3000, 9000)
        truth = [rng.randrange(1, 5) for _ in range(length)]
        start = 0
        k = 0
        while start < length - 100:
            span = rng.choice([40, 400, 900, 2300])
            end = min(length, start + span)
            noisy = (k % noisy_every) == noisy_every - 1
            keys = []
            for p in range(start, end):
                keys.append((p, 0, 0))
                if rng.random() < 0.1:
                    keys.append((p, 1, rng.randrange(2)))
            labels = {}
            for (p, i, s) in keys:
                if noisy:
                    labels[(p, i, s)] = (rng.randrange(5), rng.randrange(3))
                elif i == 0:
                    labels[(p, i, s)] = (truth[p], 1)
                else:
                    labels[(p, i, s)] = (0, 0)
            if k % 11 == 5:                                   # an empty region: every label a gap
                labels = {q: (0, 0) for q in keys}
            n_img = max(1, -(-len(keys) // 990))
            ids = rng.sample([0, 1, 2, 3, 10, 11, 12], n_img) if n_img <= 7 else list(range(n_img))
            imgs = []
            for j, cid in enumerate(ids):
                lo = 0 if j == 0 else max(0, len(keys) * j // n_img - 6)
                hi = len(keys) if j == n_img - 1 else len(keys) * (j + 1) // n_img
                rows = keys[lo:hi]
                pos = np.full((1000, 3), -1, np.int64)
                b = np.zeros(1000, np.uint8)
                r = np.zeros(1000, np.uint8)
                pos[:len(rows)] = rows
                b[:len(rows)] = [labels[q][0] for q in rows]
                r[:len(rows)] = [labels[q][1] for q in rows]
                if j > 0:                                     # the repeated keys come with other labels
                    b[:6] = rng.randrange(5)
                    r[:6] = rng.randrange(3)
                b[len(rows):] = rng.randrange(5)              # what the network calls on padding rows
                r[len(rows):] = rng.randrange(3)
                imgs.append((contig, start, end, cid, pos, b, r))
            if rng.random() < 0.2:                            # an image written twice (other labels): the second is skipped
                dup = imgs[0]
                imgs.append(dup[:5] + (np.full(1000, 3, np.uint8), np.full(1000, 2, np.uint8)))
            out.append(imgs)
            step = span - rng.randrange(150, 250) if span > 400 else span - 20
            if c == 1 and k == 3:
                step = span + 500                             # a hole
            start += max(10, step)
            k += 1
    # images in writing order, region after region ... except two regions whose images are interleaved with the next one's
    order = []
    for g, imgs in enumerate(out):
        if g % 9 == 4 and len(imgs) > 1 and g + 1 < len(out):
            order.extend(imgs[:1])
            order.extend(out[g + 1][:1])
            order.extend(imgs[1:])
            out[g + 1] = out[g + 1][1:]
        else:
            order.extend(imgs)
    return [w for w in order if w is not None]


def _arrays(windows):
    n = len(windows)
    contigs = native_io.pack_contigs([w[0] for w in windows])
    contigs[:, 200:] = 7                                      # stale bytes behind the names, as a recycled slot has them
    for i, w in enumerate(windows):
        contigs[i, len(w[0].encode())] = 0
    meta = np.array([[w[1], w[2], w[3]] for w in windows], np.int64).reshape(n, 3)
    pos = np.stack([w[4] for w in windows]) if n else np.zeros((0, 1000, 3), np.int64)
    b = np.stack([w[5] for w in windows]) if n else np.zeros((0, 1000), np.uint8)
    r = np.stack([w[6] for w in windows]) if n else np.zeros((0, 1000), np.uint8)
    return contigs, meta, np.ascontiguousarray(pos), np.ascontiguousarray(b), np.ascontiguousarray(r)


@pytest.mark.parametrize("seed,batch,threads", [(1, 1, 1), (2, 3, 3), (3, 7, 2), (4, 64, 3), (5, 100000, 1), (6, 5, 8)])
def test_streamed_stitch_equals_the_two_phase_stitch(tmp_path, seed, batch, threads, capfd):
    """Two prediction files (two ranks) written batch by batch through DataStore while two RegionStreams are fed the same
    arrays; finish_stitch must write the FASTA perform_stitch writes from the files."""
    from helen_amd.data_store import DataStore
    from helen_amd.stitch import perform_stitch
    from helen_amd.stitch_stream import RegionStream, StreamResult, finish_stitch
    rng = random.Random(seed)
    pred = tmp_path / "pred"
    pred.mkdir()
    results = []
    for rank in range(2):
        windows = _windows(rng)
        path = str(pred / ("p_%d.hdf" % rank))
        store = DataStore(path, "w")
        stream = RegionStream(path, threads=threads)
        for lo in range(0, len(windows), batch):
            contigs, meta, pos, b, r = _arrays(windows[lo:lo + batch])
            store.write_batch(contigs, meta, pos, b, r)
            stream.feed(contigs, meta, pos, b, r)
        store.close()
        res = stream.finish()
        if rank == 1:                                         # as a spawned rank hands it over: through a file
            res = StreamResult.load(res.save(str(tmp_path)))
        results.append(res)
    want = perform_stitch(str(pred), str(tmp_path / "two_phase"), "asm", threads)
    got = finish_stitch(results, str(pred), str(tmp_path / "streamed"), "asm", threads)
    a, b = open(want, "rb").read(), open(got, "rb").read()
    assert a == b and len(a) > 5000
    # ... and with every contig taken through the general procedure (no assembly from region slices)
    slow = finish_stitch(results, str(pred), str(tmp_path / "streamed_general"), "asm", threads, fast=False)
    assert open(slow, "rb").read() == a
    err = capfd.readouterr().err
    from_file = sum(r.stats["from_file"] for r in results)
    assert "JOINS FROM THE TABLE" in err
    print("seed %d batch %d: %d regions streamed, %d read back from the files, %s"
          % (seed, batch, sum(r.stats["regions"] for r in results), from_file, err.strip().splitlines()[-1]))


def test_decode_regions_equals_the_file_decode(tmp_path):
    """helen_io_decode_regions (labels in memory) against helen_io_region_sequence (the prediction file) region by region,
    including padding rows (uint32-wrapped keys), repeated keys and chunk ids in string order."""
    from helen_amd.data_store import DataStore
    rng = random.Random(77)
    windows = [w for w in _windows(rng, n_contigs=2)]
    contigs, meta, pos, b, r = _arrays(windows)
    path = str(tmp_path / "p.hdf")
    with DataStore(path, "w") as store:
        store.write_batch(contigs, meta, pos, b, r)
    regions = {}
    for i, w in enumerate(windows):
        regions.setdefault(w[:3], {}).setdefault(w[3], i)      # the first image of a chunk id is the stored one
    firsts, rows, keys = [0], [], []
    for key, ids in regions.items():
        rows.extend(ids[c] for c in sorted(ids, key=str))
        firsts.append(len(rows))
        keys.append(key)
    blob, off = native_io.decode_regions(firsts, rows, pos, b, r, threads=3)
    blob = blob.tobytes()
    for k, (contig, start, end) in enumerate(keys):
        want = native_io.region_sequence(path, contig, "%s-%d-%d" % (contig, start, end), as_bytes=True)
        assert blob[off[k]:off[k + 1]] == want, (contig, start, end)


def test_join_batch_equals_single_alignments():
    """helen_ssw_join_batch = helen_ssw_align + get_confident_positions, join by join (matching, shifted, unrelated and
    empty sides)."""
    from helen_amd.stitch import StitchOptions as O, get_confident_positions
    rng = random.Random(5)
    jobs = []
    for k in range(60):
        a = bytes(rng.choice(b"ACGT") for _ in range(rng.randrange(1, 400)))
        kind = k % 4
        if kind == 0:
            b = a
        elif kind == 1:
            cut = rng.randrange(0, max(1, len(a) // 2))
            b = bytearray(a[cut:] + bytes(rng.choice(b"ACGT") for _ in range(cut)))
            for _ in range(rng.randrange(0, 6)):
                if b:
                    b[rng.randrange(len(b))] = rng.choice(b"ACGT")
            b = bytes(b)
        elif kind == 2:
            b = bytes(rng.choice(b"ACGT") for _ in range(rng.randrange(1, 400)))
        else:
            b = b"" if k % 8 == 3 else a[:7]
        jobs.append((a, b))
    blob = b"".join(x for j in jobs for x in j)
    l_off, l_len, r_off, r_len, at = [], [], [], [], 0
    for a, b in jobs:
        l_off.append(at); l_len.append(len(a)); at += len(a)
        r_off.append(at); r_len.append(len(b)); at += len(b)
    out = native_io.ssw_join_batch(blob, l_off, l_len, r_off, r_len, O.MATCH_PENALTY, O.MISMATCH_PENALTY, O.GAP_PENALTY,
                                   O.GAP_EXTEND_PENALTY, O.OVERLAP_THRESHOLD)
    anchored = 0
    for (a, b), (score, pa, pb) in zip(jobs, out.tolist()):
        if not a or not b:
            assert (score, pa, pb) == (0, -1, -1)
            continue
        al = native_io.ssw_align(a, b, O.MATCH_PENALTY, O.MISMATCH_PENALTY, O.GAP_PENALTY, O.GAP_EXTEND_PENALTY)
        assert score == al.best_score
        if score:
            assert (pa, pb) == get_confident_positions(al), (a, b, al.cigar_string)
            anchored += pa >= 0
    assert anchored >= 20


@pytest.mark.parametrize("seed,threads,batch,n_streams", [(11, 1, 5, 1), (12, 4, 64, 1), (13, 16, 3, 1), (14, 4, 9, 3)])
def test_contigs_assembled_from_region_slices(tmp_path, seed, threads, batch, n_streams, capfd):
    """Ordinary assemblies -- regions of 700-1,200 called bases that agree with their neighbours over 100-300 of them, a
    few mismatches in the overlaps, one hole -- are put together from slices of the regions' sequences (every join of a
    contig already in the pair table, helen_amd.stitch_stream._joined_from_slices) instead of join by join: the FASTA
    must be perform_stitch's, for any number of stitch threads (= any cut of a contig into runs), and the report must say
    that the short way was taken."""
    from helen_amd.data_store import DataStore
    from helen_amd.stitch import perform_stitch
    from helen_amd.stitch_stream import RegionStream, finish_stitch
    rng = random.Random(seed)
    pred = tmp_path / "pred"
    pred.mkdir()
    windows = []
    for c in range(4):
        contig = "chr%d" % c
        length = rng.randrange(9000, 16000)
        truth = [rng.randrange(1, 5) for _ in range(length)]
        start, k = 0, 0
        while start < length - 300:
            end = min(length, start + rng.randrange(700, 1000))
            pos = np.full((1000, 3), -1, np.int64)
            n = end - start
            pos[:n, 0] = np.arange(start, end)
            pos[:n, 1:] = 0
            b = np.zeros(1000, np.uint8)
            r = np.zeros(1000, np.uint8)
            b[:n] = truth[start:end]
            r[:n] = [1 + (rng.random() < 0.2) for _ in range(n)]
            for _ in range(3):                               # a few disagreements with the neighbour
                b[rng.randrange(n)] = rng.randrange(1, 5)
            windows.append((contig, start, end, 0, pos, b, r))
            step = (end - start) - rng.randrange(100, 300)
            if c == 2 and k == 4:
                step = (end - start) + 40                     # a hole: ten Ns and a warning
            start += step
            k += 1
    # one stream per rank; with several, a contig's regions are dealt to them in turn (MarginPolish writes a contig's
    # regions into whichever thread's file, and the files are dealt to the ranks): most neighbours never meet in a stream
    results = []
    for k in range(n_streams):
        mine = windows[k::n_streams]
        path = str(pred / ("p_%d.hdf" % k))
        store = DataStore(path, "w")
        stream = RegionStream(path, threads=2)
        for lo in range(0, len(mine), batch):
            contigs, meta, pos, b, r = _arrays(mine[lo:lo + batch])
            store.write_batch(contigs, meta, pos, b, r)
            stream.feed(contigs, meta, pos, b, r)
        store.close()
        results.append(stream.finish())
    if n_streams == 1:
        assert len(results[0].pair_joins) >= len(windows) - 4 - 1
    want = perform_stitch(str(pred), str(tmp_path / "two_phase"), "asm", threads)
    capfd.readouterr()
    got = finish_stitch(results, str(pred), str(tmp_path / "streamed"), "asm", threads)
    err = capfd.readouterr().err
    a = open(want, "rb").read()
    assert open(got, "rb").read() == a and a.count(b">") == 4 and b"N" * 10 in a
    assert "4 OF 4 CONTIG(S) ASSEMBLED FROM REGION SLICES" in err and "0 ALIGNED NOW" in err, err[-400:]
    assert err.count("WARNING: NO OVERLAP IN CHUNKS") == 1
    assert ("BETWEEN REGIONS OF DIFFERENT STREAMS ALIGNED ON %d THREAD(S) FIRST" % threads in err) == (n_streams > 1)


@pytest.mark.parametrize("seed", range(40))
def test_region_slices_fuzz(tmp_path, seed):
    """The short way of finish_stitch on assemblies that are ALMOST ordinary: region lengths from 12 bases to 1,000,
    overlaps from 5 bases to longer than the shorter neighbour (nested regions), overlaps that disagree in a fifth of
    their bases (alignments without an anchor), empty regions, holes.  Whatever mixture of contigs goes the short way
    and the general way, for 1, 3 and 16 stitch threads the FASTA is perform_stitch's."""
    from helen_amd.data_store import DataStore
    from helen_amd.stitch import perform_stitch
    from helen_amd.stitch_stream import RegionStream, finish_stitch
    rng = random.Random(1000 + seed)
    pred = tmp_path / "pred"
    pred.mkdir()
    path = str(pred / "p_0.hdf")
    store = DataStore(path, "w")
    stream = RegionStream(path, threads=2)
    windows = []
    for c in range(rng.randrange(2, 6)):
        contig = "c%d" % c
        length = rng.randrange(1500, 6000)
        truth = [rng.randrange(1, 5) for _ in range(length)]
        p_odd = rng.choice([0.0, 0.0, 0.03, 0.15])             # how often a region of this contig is not ordinary
        start = 0
        while start < length - 50:
            odd = rng.random() < p_odd
            span = rng.choice([12, 60, 300, 700, 1000]) if odd else rng.randrange(500, 1000)
            end = min(length, start + span)
            n = end - start
            pos = np.full((1000, 3), -1, np.int64)
            pos[:n, 0] = np.arange(start, end)
            pos[:n, 1:] = 0
            b = np.zeros(1000, np.uint8)
            r = np.zeros(1000, np.uint8)
            b[:n] = truth[start:end]
            r[:n] = [1 + (rng.random() < 0.3) for _ in range(n)]
            if odd and rng.random() < 0.3:
                for j in range(n):                             # a noisy region: its overlaps will not anchor
                    if rng.random() < 0.2:
                        b[j] = rng.randrange(1, 5)
            elif rng.random() < 0.5:
                for _ in range(rng.randrange(1, 6)):           # a few disagreements with the neighbours
                    b[rng.randrange(n)] = rng.randrange(1, 5)
            if odd and rng.random() < 0.1:
                b[:n] = 0                                      # an empty region
                r[:n] = 0
            windows.append((contig, start, end, 0, pos, b, r))
            if end == length:
                break
            ov = rng.choice([5, 20, 150, 400, 1200]) if odd else rng.randrange(60, 300)
            step = n - ov
            if odd and rng.random() < 0.2:
                step = n + rng.randrange(1, 200)               # a hole
            start += max(3, step)
    for lo in range(0, len(windows), 7):
        contigs, meta, pos, b, r = _arrays(windows[lo:lo + 7])
        store.write_batch(contigs, meta, pos, b, r)
        stream.feed(contigs, meta, pos, b, r)
    store.close()
    res = stream.finish()
    for threads in (1, 3, 16):
        want = perform_stitch(str(pred), str(tmp_path / ("two_phase%d" % threads)), "asm", threads)
        got = finish_stitch([res], str(pred), str(tmp_path / ("streamed%d" % threads)), "asm", threads)
        assert open(got, "rb").read() == open(want, "rb").read(), (seed, threads)
