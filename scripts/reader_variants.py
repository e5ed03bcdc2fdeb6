"""Reader rate per storage layout (no GPU work): for every way `helen_amd.synthetic.write_image_file` can store an image
file -- contiguous, chunked, gzip 1 / 4 / 9, shuffle, fletcher32, libver=latest -- windows/s of the product's reader
(helen_io_read_image_runs: native threads over the direct scanner) with 1, 4 and 8 threads, and of libhdf5 reading the
same file (one thread: the library is serialised), pileup-like AND uniform-random pixel values (deflate's speed depends
on the data).  helen_amd.host_plan reads its per-layout rates off this table (profiles/r04_reader_variants.txt).

    python scripts/reader_variants.py [--windows 4096] [--out profiles/r04_reader_variants.txt]
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

VARIANTS = [
    ("contiguous", {}),
    ("chunked (100,90)", dict(chunks=(100, 90))),
    ("chunked (1000,90)", dict(chunks=(1000, 90))),
    ("gzip 1", dict(gzip=1)),
    ("gzip 4", dict(gzip=4)),
    ("gzip 9", dict(gzip=9)),
    ("shuffle + gzip 4", dict(gzip=4, shuffle=True)),
    ("fletcher32", dict(fletcher32=True)),
    ("latest", dict(libver="latest")),
    ("latest chunked (100,90)", dict(libver="latest", chunks=(100, 90))),
    ("latest gzip 4", dict(libver="latest", gzip=4)),
]

CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, %(root)r)
from helen_amd import native_io
path, n, threads = %(path)r, %(n)d, %(threads)d
got = native_io.index_images(path)
images = np.ones((n, 1000, 90), np.uint8); positions = np.ones((n, 1000, 3), np.int64)
meta = np.ones((n, 3), np.int64); contigs = np.ones((n, native_io.NAME_BYTES), np.uint8)
t0 = time.time()
lib = native_io.read_image_runs([(path, 0, n)], threads, images, positions, meta, contigs)
print(n / (time.time() - t0), lib, int(images.sum() %% 1000003))
'''


def rate(path, n, threads, reader):
    env = dict(os.environ)
    env.pop("HELEN_IO_READER", None)
    if reader:
        env["HELEN_IO_READER"] = reader
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                       "path": path, "n": n, "threads": threads}],
                       env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-800:])
    v, lib, digest = r.stdout.split()
    return float(v), int(lib), digest


def main():
    from helen_amd.synthetic import write_image_file
    from helen_amd.weights import make_images
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=4096)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    d = tempfile.mkdtemp(prefix="helen_rv_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    lines = ["reader windows/s per storage layout: %d windows per file, files in %s, usable CPUs %d"
             % (a.windows, os.path.dirname(d), len(os.sched_getaffinity(0))),
             "%-26s %-8s %9s | %9s %9s %9s | %9s | %s" % ("layout", "pixels", "file MB", "1 thread", "4 threads", "8 threads",
                                                        "libhdf5", "read by")]
    try:
        for mode in ("pileup", "uniform"):
            img = make_images(a.windows, seed=3, mode=mode)
            for name, kw in VARIANTS:
                if mode == "uniform" and "gzip" not in name and name != "contiguous":
                    continue          # only deflate cares what the pixels are
                path = os.path.join(d, "v.h5")
                t0 = time.time()
                write_image_file(path, img, **kw)
                tw = time.time() - t0
                rates, digests = [], set()
                for threads in (1, 4, 8):
                    v, lib, dg = rate(path, a.windows, threads, None)
                    rates.append(v)
                    digests.add(dg)
                vl, _, dg = rate(path, a.windows, 1, "libhdf5")
                digests.add(dg)
                assert len(digests) == 1, "the readers disagree on %s" % name
                lines.append("%-26s %-8s %9.1f | %9.0f %9.0f %9.0f | %9.0f | %s   (written in %.1f s)"
                             % (name, mode, os.path.getsize(path) / 1e6, rates[0], rates[1], rates[2], vl,
                                "scanner" if lib == 0 else "libhdf5 (%d of %d)" % (lib, a.windows), tw))
                print(lines[-1], flush=True)
                os.unlink(path)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    text = "\n".join(lines) + "\n"
    if a.out:
        open(a.out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
