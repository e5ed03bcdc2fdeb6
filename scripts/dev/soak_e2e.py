"""Developer probe (GPU box): call_consensus N times over the same synthetic directory; every run's prediction file
must hold byte-identical labels (the path is deterministic) -- hunts for rare races in the reader / device / writer
pipeline.    python scripts/dev/soak_e2e.py [windows] [runs]"""
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from helen_amd.model_handler import ModelHandler  # noqa: E402
from helen_amd.synthetic import write_image_dir  # noqa: E402
from helen_amd.weights import make_weights  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = tempfile.mkdtemp(prefix="helen_soak_", dir="/dev/shm")
try:
    model = os.path.join(d, "model.pkl")
    ModelHandler.save_model(make_weights(seed=20260928, head_scale=8.0, input_scale=1 / 64.0), None, 128, 1, 0, model)
    write_image_dir(os.path.join(d, "img"), n, n_files=8, direct=True, short_every=97)
    digests = []
    for k in range(runs):
        out = os.path.join(d, "out%d" % k)
        t0 = time.time()
        r = subprocess.run([sys.executable, "-m", "helen_amd", "call_consensus", "-i", os.path.join(d, "img"), "-m", model,
                            "-b", "256", "-w", "8", "-o", out, "-p", "p", "-g"], cwd=ROOT, capture_output=True, text=True)
        if r.returncode != 0:
            print("run %d FAILED rc %d\n%s" % (k, r.returncode, r.stderr[-3000:]))
            sys.exit(1)
        h = hashlib.sha1()
        for root, _, files in os.walk(out):
            for f in sorted(files):
                if f.endswith("hdf"):
                    with open(os.path.join(root, f), "rb") as fh:
                        for block in iter(lambda: fh.read(1 << 24), b""):
                            h.update(block)
        digests.append(h.hexdigest())
        print("run %d: %.1f s  %s" % (k, time.time() - t0, digests[-1][:16]), flush=True)
        shutil.rmtree(out, ignore_errors=True)
    print("%d runs, %d distinct outputs" % (runs, len(set(digests))))
    sys.exit(0 if len(set(digests)) == 1 else 2)
finally:
    shutil.rmtree(d, ignore_errors=True)
