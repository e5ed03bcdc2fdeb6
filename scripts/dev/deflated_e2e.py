"""Developer probe (GPU box): `bin/helen call_consensus -g` on DEFLATED image files (gzip 4, chunked (256, 90): what an
h5py writer with compression="gzip" stores), pileup-like pixels -- is a run on compressed MarginPolish output bound by
its readers or by the device?  Files are written by libhdf5 (1.3 k windows/s per process: eight processes).
    python scripts/dev/deflated_e2e.py [windows=65536] [reader threads "8,16"]"""
import concurrent.futures
import multiprocessing as mp
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def _write(args):
    path, n, seed, first = args
    from helen_amd.synthetic import write_image_file
    from helen_amd.weights import make_images
    write_image_file(path, make_images(n, seed=seed, mode="pileup"), contig="chr_z", first_window=first, gzip=4)
    return os.path.getsize(path)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    workers = (sys.argv[2] if len(sys.argv) > 2 else "8,16").split(",")
    from helen_amd.model_handler import ModelHandler
    from helen_amd.weights import make_weights
    d = tempfile.mkdtemp(prefix="helen_z_", dir="/dev/shm")
    try:
        img = os.path.join(d, "img")
        os.makedirs(img)
        files = 16
        per = n // files
        t0 = time.time()
        with concurrent.futures.ProcessPoolExecutor(8, mp_context=mp.get_context("spawn")) as ex:
            sizes = list(ex.map(_write, [(os.path.join(img, "z_%02d.h5" % k), per, 100 + k, k * per) for k in range(files)]))
        print("%d windows in %d deflated files (%.1f MB, %.0f %% of the raw pixels) written in %.0f s"
              % (per * files, files, sum(sizes) / 1e6, 100.0 * sum(sizes) / (per * files * 114000.0), time.time() - t0), flush=True)
        model = os.path.join(d, "model.pkl")
        ModelHandler.save_model(make_weights(input_scale=1.0 / 64.0), None, 128, 1, 0, model)
        for w in workers:
            out = os.path.join(d, "out" + w)
            t0 = time.time()
            r = subprocess.run([os.path.join(ROOT, "bin", "helen"), "call_consensus", "-i", img, "-m", model, "-b", "256", "-w", w,
                                "-t", "16", "-o", out, "-p", "p", "-g"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            dt = time.time() - t0
            info = [ln for ln in r.stderr.splitlines() if "WINDOWS IN" in ln or "HOST PLAN" in ln or "HOST-BOUND" in ln]
            print("-w %s rc %d: %.2f s = %.0f windows/s" % (w, r.returncode, dt, per * files / dt))
            print("\n".join("    " + ln[:260] for ln in info[-3:]))
            if r.returncode:
                print(r.stderr[-2000:])
            shutil.rmtree(out, ignore_errors=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
