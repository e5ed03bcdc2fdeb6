"""Developer probe (GPU box): `python -m helen_amd polish` on a synthetic chr20-sized directory, wall-clocked as a user
would see it (process start, imports, call_consensus, stitch).  Random weights give random labels, so the overlaps of
neighbouring regions do not agree and stitch mostly inserts fillers: a plumbing / scale check, not a stitch benchmark
(scripts/stitch_bench.py is that)."""
import os
import shutil
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from helen_amd.model_handler import ModelHandler  # noqa: E402
from helen_amd.synthetic import write_image_dir  # noqa: E402
from helen_amd.weights import make_weights  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
threads = sys.argv[2] if len(sys.argv) > 2 else "16"
d = tempfile.mkdtemp(prefix="helen_polish_", dir="/dev/shm")
try:
    model = os.path.join(d, "model.pkl")
    ModelHandler.save_model(make_weights(seed=20260928, head_scale=8.0, input_scale=1 / 64.0), None, 128, 1, 0, model)
    t0 = time.time()
    write_image_dir(os.path.join(d, "img"), n, n_files=16, direct=True)
    print("inputs written in %.1f s" % (time.time() - t0), flush=True)
    t0 = time.time()
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([os.path.join(root, "bin", "helen"), "polish", "-i", os.path.join(d, "img"), "-m", model, "-b", "256",
                        "-w", "8", "-t", threads, "-o", os.path.join(d, "out"), "-p", "asm", "-g"], cwd=root,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    dt = time.time() - t0
    info = [l for l in r.stderr.splitlines() if l.startswith("INFO") and ("WINDOWS IN" in l or "POLISHED" in l or "STITCH" in l)]
    print("rc", r.returncode, "polish wall %.2f s = %.0f windows/s" % (dt, n / dt))
    print("\n".join(info[-4:]))
    print("stderr lines:", len(r.stderr.splitlines()), " outputs:", sorted(os.listdir(os.path.join(d, "out"))))
    if r.returncode != 0:
        print(r.stderr[-3000:])
finally:
    shutil.rmtree(d, ignore_errors=True)
