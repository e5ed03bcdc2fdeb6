"""Stitch behind a multi-rank run (helen_amd/stitch_collect.py): the ranks export their decoded regions to collector
processes sharded by contig; the FASTA must be the one `perform_stitch` writes from the finished prediction files --
whatever the number of ranks, collectors and threads, and however the regions of a contig are dealt over the ranks."""
import os
import time
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from helen_amd import native_io  # noqa: E402
from test_stitch_stream import _arrays, _windows  # noqa: E402

pytestmark = pytest.mark.skipif(not native_io.available(), reason="libhelen_io.so is not built")


def test_records_survive_any_cut_of_the_byte_stream(tmp_path):
    """What a rank appends, a collector reads back region for region -- also when its reads end in the middle of a header,
    a name or a sequence, and the end marker arrives on its own."""
    from helen_amd import stitch_collect as sc
    prefix = str(tmp_path / "x")
    open(sc._path(prefix, 0, 0), "wb").close()
    exp = sc.RegionExport(prefix, 0, 1)
    rng = random.Random(3)
    want = []
    for k in range(200):
        key = ("contig/%d.é" % (k % 5), k * 100, k * 100 + rng.randrange(1, 5000))
        seq = bytes(rng.choice(b"ACGT") for _ in range(rng.randrange(0, 3000)))
        if k % 17 == 3:
            exp.from_file(key)
            want.append((key, None))
        else:
            exp.write([key], [seq])
            want.append((key, seq))
    exp.close()
    data = open(sc._path(prefix, 0, 0), "rb").read()
    # replay the bytes in random pieces through a follower reading another file
    replay = sc._path(prefix, 1, 0)
    open(replay, "wb").close()
    f = sc._Follower(replay)
    got, at = [], 0
    with open(replay, "ab", buffering=0) as out:
        while at < len(data):
            n = rng.choice([1, 3, 20, 21, 700, 5000])
            out.write(data[at:at + n])
            at += n
            records, _ = f.poll()
            got.extend(records)
    assert f.ended and got == want
    f.close()


def _deal(windows, ranks, how, rng):
    """Regions (runs of images of one (contig, start, end)) dealt to `ranks` lists: "blocks" = eight consecutive regions
    to a rank, "scatter" = every region to a random rank, "contigs" = a whole contig to one rank."""
    groups = []
    for w in windows:
        key = (w[0], w[1], w[2])
        if groups and groups[-1][0] == key:
            groups[-1][1].append(w)
        else:
            groups.append((key, [w]))
    out = [[] for _ in range(ranks)]
    for i, (key, imgs) in enumerate(groups):
        if how == "blocks":
            r = (i // 8) % ranks
        elif how == "scatter":
            r = rng.randrange(ranks)
        else:
            r = int(key[0][3]) % ranks                          # ("ctg<k>...": contig k)
        out[r].extend(imgs)
    return out


@pytest.mark.parametrize("seed,ranks,threads,how", [(1, 2, 1, "blocks"), (2, 3, 4, "scatter"), (3, 2, 16, "scatter"),
                                                    (4, 4, 9, "contigs"), (5, 3, 32, "blocks"), (6, 11, 8, "scatter")])
def test_collectors_write_the_fasta_of_the_two_phase_stitch(tmp_path, seed, ranks, threads, how, capfd):
    from helen_amd.data_store import DataStore
    from helen_amd.stitch import perform_stitch
    from helen_amd.stitch_collect import CollectorRun, RegionExport, collectors_for
    from helen_amd.stitch_stream import RegionStream
    rng = random.Random(seed)
    pred = tmp_path / "pred"
    pred.mkdir()
    # (eleven ranks: p_10.hdf sorts before p_2.hdf and os.listdir sorts nothing -- between regions of one span the
    # directory's listing order decides, as in perform_stitch, not the rank order)
    dealt = _deal(_windows(rng, n_contigs=5 if ranks < 10 else 8), ranks, how, rng)
    for r in range(ranks):
        assert dealt[r], "rank %d got nothing (perform_stitch refuses an empty prediction file)" % r
    files = [str(pred / ("p_%d.hdf" % r)) for r in range(ranks)]
    run = CollectorRun(files, threads, directory=str(tmp_path)).start()
    assert run.buckets == collectors_for(threads) == max(1, min(8, threads // 4))
    try:
        for r in range(ranks):
            store = DataStore(files[r], "w")
            stream = RegionStream(files[r], threads=2, export=RegionExport(*((run.export_spec()[0], r, run.export_spec()[1]))))
            for lo in range(0, len(dealt[r]), 5):
                contigs, meta, pos, b, rl = _arrays(dealt[r][lo:lo + 5])
                store.write_batch(contigs, meta, pos, b, rl)
                stream.feed(contigs, meta, pos, b, rl)
            store.close()
            res = stream.finish()
            assert res.regions == {} and res.stats.get("exported")
        got = run.finish(str(tmp_path / "streamed"), "asm")
    except BaseException:
        run.abort()
        raise
    want = perform_stitch(str(pred), str(tmp_path / "two_phase"), "asm", threads)
    a = open(want, "rb").read()
    assert open(got, "rb").read() == a and len(a) > 5000
    err = capfd.readouterr().err
    assert "STITCH COLLECTOR(S) OVER %d RANK(S)" % ranks in err
    left = [f for f in os.listdir(str(tmp_path)) if f.startswith("helen_regions_")]
    assert left == [], left                                   # record files and parts are gone


def test_a_failed_run_leaves_nothing_behind(tmp_path):
    from helen_amd.stitch_collect import CollectorRun
    files = [str(tmp_path / "p_0.hdf"), str(tmp_path / "p_1.hdf")]
    run = CollectorRun(files, 8, directory=str(tmp_path)).start()
    assert all(p.is_alive() for p in run.procs)
    run.abort()
    assert run.procs == [] and [f for f in os.listdir(str(tmp_path)) if f.startswith("helen_regions_")] == []


def test_what_a_killed_run_left_behind_is_swept(tmp_path):
    """A run's record and part files live in a directory of its own whose lock file the parent holds (flock) while it
    lives.  The next run removes only what is provably stale: nobody holds the lock AND nothing changed for ten minutes --
    a live run in another PID namespace that shares /dev/shm (containers with --ipc=host) holds its lock and is left alone."""
    import fcntl
    from helen_amd.stitch_collect import CollectorRun
    old = time.time() - 3600

    def make(name, locked, aged):
        d = tmp_path / name
        d.mkdir()
        (d / "r_0_0.bin").write_bytes(b"x")
        lock = open(str(d / "lock"), "w")
        if aged:
            for f in (d / "r_0_0.bin", d / "lock", d):
                os.utime(str(f), (old, old))
        if locked:
            fcntl.flock(lock, fcntl.LOCK_EX | fcntl.LOCK_NB)
            return lock
        lock.close()
        return None
    held = make("helen_regions_live", True, True)              # old but its parent lives
    make("helen_regions_fresh", False, False)                  # nobody holds it, but it changed a moment ago
    make("helen_regions_stale", False, True)
    other = tmp_path / "helen_slot_1_2_3"
    other.write_bytes(b"x")
    assert CollectorRun.sweep(str(tmp_path)) == 1
    assert sorted(n for n in os.listdir(str(tmp_path))) == ["helen_regions_fresh", "helen_regions_live", "helen_slot_1_2_3"]
    held.close()


def test_the_spill_goes_to_ram_only_when_it_fits(tmp_path, monkeypatch):
    """The collectors' files are RAM when they sit under /dev/shm: the directory is chosen against what the run may still
    take (free space of the tmpfs, half of the available RAM), with a floor of 1 GiB -- Docker's default 64 MB /dev/shm is
    never used; otherwise the files go beside the prediction files."""
    from helen_amd import host_plan, stitch_stream
    from helen_amd.stitch_collect import CollectorRun
    monkeypatch.setattr(host_plan, "shm_free_bytes", lambda path="/dev/shm": 64 << 20)
    monkeypatch.setattr(host_plan, "ram_available_bytes", lambda: 256 << 30)
    assert stitch_stream.spill_directory(0) is None
    files = [str(tmp_path / "pred" / "p_0.hdf"), str(tmp_path / "pred" / "p_1.hdf")]
    os.makedirs(str(tmp_path / "pred"))
    run = CollectorRun(files, 4, expected_bytes=10 << 20)
    try:
        assert os.path.dirname(run.directory) == str(tmp_path / "pred")
    finally:
        run.abort()
    assert os.listdir(str(tmp_path / "pred")) == []
    monkeypatch.setattr(host_plan, "shm_free_bytes", lambda path="/dev/shm": 200 << 30)
    if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK):
        assert stitch_stream.spill_directory(6 << 30) == "/dev/shm"
        assert stitch_stream.spill_directory(120 << 30) is None         # more than half of the available RAM


def test_consumed_records_are_given_back(tmp_path):
    """A collector punches the pages of what it has read out of a rank's record file (tmpfs pages are RAM): the file keeps
    its size, its blocks go."""
    from helen_amd.stitch_collect import _Follower, _HEADER
    path = str(tmp_path / "r_0_0.bin")
    rec = _HEADER.pack(4, 0, 1000, 1000) + b"ctgA" + b"A" * 1000
    n = (96 << 20) // len(rec)
    with open(path, "wb") as f:
        f.write(rec * n)
    before = os.stat(path).st_blocks
    fo = _Follower(path)
    total = 0
    while True:
        records, got = fo.poll()
        if not got:
            break
        total += len(records)
        assert all(k == ("ctgA", 0, 1000) and len(sq) == 1000 for k, sq in records)
    fo.close()
    assert total == n
    after = os.stat(path)
    assert after.st_size == n * len(rec)
    if fo.punched:                                           # (a file system without hole punching: nothing to check)
        assert after.st_blocks < before // 2, (before, after.st_blocks)
