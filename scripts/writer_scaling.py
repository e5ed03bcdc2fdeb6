"""Writer scaling probe: N processes each write their share of synthetic label rows to their own
prediction file (no GPU work).    python scripts/writer_scaling.py --windows 65536 --writers 1,2,4,8,16"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def work(k, nw, n, out, q):
    from helen_amd import native_io
    rng = np.random.default_rng(k)
    B = 4096
    names = native_io.pack_contigs(["contig_0"] * B)
    positions = np.zeros((B, 1000, 3), dtype=np.int64)
    bases = rng.integers(0, 5, (B, 1000), dtype=np.uint8)
    rles = rng.integers(0, 11, (B, 1000), dtype=np.uint8)
    w = native_io.Writer(out + "_%d.hdf" % k)
    q.put("ready")
    t0 = time.time()
    done = 0
    for s in range(0, n, B):
        meta = np.zeros((B, 3), dtype=np.int64)
        idx = np.arange(s, s + B)
        meta[:, 0] = idx * 800
        meta[:, 1] = idx * 800 + 1000
        sel = np.nonzero(idx % nw == k)[0].astype(np.int32)
        w.write(names, meta, positions, bases, rles, sel=sel)
        done += sel.size
    w.close()
    q.put((done, time.time() - t0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=65536)
    ap.add_argument("--writers", default="1,2,4,8")
    ap.add_argument("--out", default="/dev/shm/helen_ws")
    a = ap.parse_args()
    ctx = mp.get_context("spawn")
    for nw in [int(x) for x in a.writers.split(",")]:
        q = ctx.Queue()
        ps = [ctx.Process(target=work, args=(k, nw, a.windows, a.out, q)) for k in range(nw)]
        for p in ps:
            p.start()
        res = []
        while len(res) < nw:
            r = q.get()
            if r != "ready":
                res.append(r)
        for p in ps:
            p.join()
        n = sum(r[0] for r in res)
        dt = max(r[1] for r in res)
        print("%2d writers: %d windows in %.2f s = %.0f windows/s (%.0f per writer)" % (nw, n, dt, n / dt, n / dt / nw),
              flush=True)
        for k in range(nw):
            os.unlink(a.out + "_%d.hdf" % k)


if __name__ == "__main__":
    main()
