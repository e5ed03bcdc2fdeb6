"""Stitch throughput probe (CPU): synthetic prediction files cut from a random sequence -> FASTA.

    python scripts/stitch_bench.py --windows 50000 --threads 8
"""
import argparse
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from helen_amd.data_store import DataStore
    from helen_amd.stitch import perform_stitch
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=50000)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--files", type=int, default=4)
    ap.add_argument("--contigs", type=int, default=1, help="cut the sequence into this many contigs (an assembly "
                                                              "is thousands of contigs, most of them short)")
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    n = a.windows
    total = 800 * n + 1000
    seq_b = rng.integers(1, 5, total, dtype=np.uint8)       # bases A..T
    seq_r = rng.integers(1, 4, total, dtype=np.uint8)       # run lengths 1..3
    d = tempfile.mkdtemp(prefix="helen_sb_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        t0 = time.time()
        stores = [DataStore(os.path.join(d, "p_%d.hdf" % k), "w") for k in range(a.files)]
        B = 2048
        pos = np.zeros((B, 1000, 3), np.int64)
        for s in range(0, n, B):
            e = min(n, s + B)
            m = e - s
            starts = np.arange(s, e) * 800
            meta = np.stack([starts, starts + 1000, np.zeros(m, np.int64)], axis=1)
            pos[:m, :, 0] = starts[:, None] + np.arange(1000)[None, :]
            idx = starts[:, None] + np.arange(1000)[None, :]
            per = -(-n // a.contigs)                              # windows per contig
            names = ["contig_%05d" % (w // per) for w in range(s, e)]
            stores[(s // B) % a.files].write_batch(names, meta, pos[:m], seq_b[idx], seq_r[idx])
        for st in stores:
            st.close()
        t1 = time.time()
        out = perform_stitch(d, os.path.join(d, "out"), "asm", a.threads)
        t2 = time.time()
        length = sum(len(line) for line in open(out).read().split("\n")[1::2])
        print("wrote %d windows (%d contigs) in %.1f s; stitched in %.2f s with %d threads = %.0f windows/s, %.1f Mbase/s "
              "(FASTA %d bases)" % (n, a.contigs, t1 - t0, t2 - t1, a.threads, n / (t2 - t1), length / (t2 - t1) / 1e6,
                                    length))
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
