"""Developer probe (CPU): what `finish_stitch` -- the part of `helen polish` that runs AFTER the last window -- costs on a
stream of N regions (regions of ~4 kb cut from random contigs with 200-base overlaps, as the simulated assembly gives),
with the joins speculated as in a run.      python scripts/dev/finish_stitch_bench.py [regions=100000] [threads=16] [profile]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from helen_amd import stitch_stream  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    rng = np.random.default_rng(5)
    contigs = 32
    per = n // contigs
    d = tempfile.mkdtemp(prefix="fsb_", dir="/dev/shm")
    pred = os.path.join(d, "pred")
    os.makedirs(pred)
    f = os.path.join(pred, "p_0.hdf")
    open(f, "wb").close()
    stream = stitch_stream.RegionStream(f, threads)
    t0 = time.time()
    for c in range(contigs):
        truth = rng.integers(0, 4, per * 3300 + 1000).astype(np.uint8)
        truth = np.frombuffer(b"ACGT", np.uint8)[truth]
        keys, parts, off = [], [], [0]
        for r in range(per):
            s = r * 3300
            keys.append(("contig_%d" % c, s, s + 4000))
            parts.append(truth[s:s + 4000])
            off.append(off[-1] + 4000)
            if len(keys) == 1365:
                stream._accept(keys, np.concatenate(parts), np.array(off))
                keys, parts, off = [], [], [0]
        if keys:
            stream._accept(keys, np.concatenate(parts), np.array(off))
    res = stream.finish()
    t1 = time.time()
    print("stream of %d regions built and joined in %.2f s (%d joins)" % (len(res.regions), t1 - t0, len(res.joins)))
    if len(sys.argv) > 3:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        stitch_stream.finish_stitch([res], pred, d, "asm", threads)
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    else:
        for _ in range(3):
            t1 = time.time()
            stitch_stream.finish_stitch([res], pred, d, "asm", threads)
            print("finish_stitch: %.3f s, FASTA %d bytes" % (time.time() - t1, os.path.getsize(os.path.join(d, "asm.fa"))))
    import shutil
    shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
