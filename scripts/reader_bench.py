"""Reader scaling probe: N worker processes fill shared slots from a synthetic image directory;
aggregate windows/s per N (no GPU work).

    python scripts/reader_bench.py --windows 65536 --workers 8,16,32,64
"""
import argparse
import concurrent.futures as cf
import multiprocessing as mp
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from helen_amd.sequence_dataset import SequenceDataset, SharedSlot, fill_shared
    from helen_amd.synthetic import write_image_dir
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=32768)
    ap.add_argument("--workers", default="8,16")
    ap.add_argument("--task", type=int, default=128, help="windows per task")
    a = ap.parse_args()
    d = tempfile.mkdtemp(prefix="helen_rb_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        t0 = time.time()
        write_image_dir(d, a.windows, n_files=16, direct=True)
        print("wrote %d windows in %.1f s" % (a.windows, time.time() - t0), flush=True)
        pairs = SequenceDataset(d).all_images
        cap = 4096
        slots = [SharedSlot(cap) for _ in range(4)]
        for nw in [int(x) for x in a.workers.split(",")]:
            with cf.ProcessPoolExecutor(nw, mp_context=mp.get_context("spawn")) as pool:
                list(pool.map(abs, range(nw * 4)))          # spawn all workers first
                warm = [pool.submit(fill_shared, slots[0].path, cap, 0, pairs[:16]) for _ in range(nw)]
                [f.result() for f in warm]
                t0 = time.time()
                futs = []
                for ci, lo in enumerate(range(0, len(pairs), cap)):
                    sl = slots[ci % len(slots)]
                    chunk = pairs[lo:lo + cap]
                    for off in range(0, len(chunk), a.task):
                        futs.append(pool.submit(fill_shared, sl.path, cap, off, chunk[off:off + a.task]))
                n = sum(f.result() for f in futs)
                dt = time.time() - t0
                print("%3d workers: %d windows in %.2f s = %.0f windows/s (%.0f per worker)"
                      % (nw, n, dt, n / dt, n / dt / nw), flush=True)
        for sl in slots:
            sl.close()
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
