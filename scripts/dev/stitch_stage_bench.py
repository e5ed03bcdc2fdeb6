#!/usr/bin/env python3
"""Developer probe (any machine): the stitch stage of `polish` alone -- helen_amd.stitch_stream.RegionStream fed slots of
4,096 windows of a simulated assembly with PERFECT labels (the truth's own base / run length per row), as predict()'s
stitch thread feeds it the label buffers of a device call; windows/s of feed(), of finish(), and where feed()'s time goes.
    python scripts/dev/stitch_stage_bench.py [windows=60000] [threads=16] [slot=4096]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from helen_amd import native_io  # noqa: E402
from helen_amd.options import ImageSizeOptions  # noqa: E402
from helen_amd.stitch_stream import RegionStream  # noqa: E402
from helen_amd.synthetic import SimContig, assembly_spec  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    slot = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    L = ImageSizeOptions.SEQ_LENGTH
    spec = assembly_spec(n, 4)
    contigs, meta, positions, bases, rles = [], [], [], [], []
    for k, (name, npos, kw) in enumerate(spec):
        c = SimContig(name, npos, 20260929 + k, **kw)
        wins = c.windows
        first = np.array([w[2] for w in wins], np.int64)
        lengths = np.array([w[3] - w[2] for w in wins], np.int64)
        row = first[:, None] + np.arange(L, dtype=np.int64)[None, :]
        live = np.arange(L)[None, :] < lengths[:, None]
        row = np.where(live, row, 0)
        positions.append(np.where(live[:, :, None], c.position[row], -1))
        bases.append(np.where(live, c.label_base[row], 0).astype(np.uint8))
        rles.append(np.where(live, c.label_rle[row], 0).astype(np.uint8))
        m = np.zeros((len(wins), 3), np.int64)
        m[:, 0] = [c.regions[w[0]][0] for w in wins]
        m[:, 1] = [c.regions[w[0]][1] for w in wins]
        m[:, 2] = [w[1] for w in wins]
        meta.append(m)
        contigs += [name] * len(wins)
    meta, positions, bases, rles = (np.concatenate(x) for x in (meta, positions, bases, rles))
    names = native_io.pack_contigs(contigs)
    total = meta.shape[0]
    h0 = native_io.ssw_fast_path_counts()
    import resource
    stream = RegionStream("/nonexistent/p_0.hdf", threads)
    c0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.time()
    for lo in range(0, total, slot):
        hi = min(total, lo + slot)
        stream.feed(names[lo:hi], meta[lo:hi], positions[lo:hi], bases[lo:hi], rles[lo:hi])
    t1 = time.time()
    res = stream.finish()
    t2 = time.time()
    c1 = resource.getrusage(resource.RUSAGE_SELF)
    cpu = (c1.ru_utime + c1.ru_stime) - (c0.ru_utime + c0.ru_stime)
    h1 = native_io.ssw_fast_path_counts()
    print("%d windows, %d regions, %d threads: feed %.2f s = %.0f windows/s; finish (alignments still queued) %.2f s; "
          "aligner shortcut %d of %d" % (total, len(res.regions), threads, t1 - t0, total / (t1 - t0), t2 - t1, h1[0] - h0[0],
                                         h1[0] - h0[0] + h1[1] - h0[1]))
    print("CPU time of the stage (all its threads): %.2f s for %d windows = %.2f CPUs at 81,000 windows/s (one MI355X, fp32)"
          % (cpu, total, cpu / total * 81000.0))
    for on in (False,):        # the same joins through the three passes, for the price of an alignment without the shortcut
        native_io.ssw_fast_path(on)
        stream2 = RegionStream("/nonexistent/p_0.hdf", threads)
        c0 = resource.getrusage(resource.RUSAGE_SELF)
        for lo in range(0, total, slot):
            hi = min(total, lo + slot)
            stream2.feed(names[lo:hi], meta[lo:hi], positions[lo:hi], bases[lo:hi], rles[lo:hi])
        stream2.finish()
        c1 = resource.getrusage(resource.RUSAGE_SELF)
        cpu2 = (c1.ru_utime + c1.ru_stime) - (c0.ru_utime + c0.ru_stime)
        native_io.ssw_fast_path(True)
        print("with the aligner's shortcuts off: %.2f s = %.2f CPUs at 81,000 windows/s" % (cpu2, cpu2 / total * 81000.0))
    print("stream seconds:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in stream.seconds.items()})


if __name__ == "__main__":
    main()
