"""Prediction-writer throughput (windows/s) of libhelen_io.so on synthetic label rows.

    python scripts/writer_bench.py [--windows 20000] [--out /dev/shm/wb.hdf]
"""
import argparse
import os
import time

import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from helen_amd import native_io


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=20000)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--per-region", type=int, default=3, help="chunk ids (images) per region")
    ap.add_argument("--out", default="/dev/shm/helen_writer_bench.hdf")
    a = ap.parse_args()
    n = a.windows
    rng = np.random.default_rng(0)
    names = ["contig_%d" % (i // 5000) for i in range(n)]
    meta = np.zeros((n, 3), dtype=np.int64)
    region = np.arange(n) // a.per_region
    meta[:, 0] = region * 2400
    meta[:, 1] = region * 2400 + 2400
    meta[:, 2] = np.arange(n) % a.per_region
    positions = np.zeros((a.batch, 1000, 3), dtype=np.int64)
    positions[:, :, 0] = np.arange(1000)[None, :]
    bases = rng.integers(0, 5, (a.batch, 1000), dtype=np.uint8)
    rles = rng.integers(0, 11, (a.batch, 1000), dtype=np.uint8)
    if os.path.exists(a.out):
        os.unlink(a.out)
    t0 = time.time()
    w = native_io.Writer(a.out)
    for s in range(0, n, a.batch):
        e = min(n, s + a.batch)
        w.write(native_io.pack_contigs(names[s:e]), meta[s:e], positions[:e - s], bases[:e - s], rles[:e - s])
    t1 = time.time()
    w.close()
    t2 = time.time()
    print("%d windows: write %.2f s, close %.2f s -> %.0f windows/s, file %.1f MB"
          % (n, t1 - t0, t2 - t1, n / (t2 - t0), os.path.getsize(a.out) / 1e6))
    os.unlink(a.out)


if __name__ == "__main__":
    main()
