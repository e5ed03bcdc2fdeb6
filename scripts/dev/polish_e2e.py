"""Developer probe (GPU box): `bin/helen polish -g` on a simulated assembly of ~N images (helen_amd.synthetic, the trained
network of tests/golden/trained_synth.npz), wall-clocked as a user sees it: process start, imports, device context,
call_consensus, stitch (pipelined behind the inference; HELEN_STITCH_PIPELINE=0 for the two phases of round 4).
    python scripts/dev/polish_e2e.py [N=300000] [threads=16] [repeats=2] [extra arguments of the command, e.g. "-d_ids 0,0"]
        [blocks=1: every contig's regions cut into that many runs dealt over the 16 files, as MarginPolish scatters them]"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from helen_amd.model_handler import ModelHandler  # noqa: E402
from helen_amd.synthetic import assembly_spec, write_assembly_dir  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    threads = sys.argv[2] if len(sys.argv) > 2 else "16"
    repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    extra = sys.argv[4].split() if len(sys.argv) > 4 else []
    blocks = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    d = tempfile.mkdtemp(prefix="helen_polish_", dir="/dev/shm")
    try:
        model = os.path.join(d, "model.pkl")
        z = np.load(os.path.join(ROOT, "tests", "golden", "trained_synth.npz"))
        ModelHandler.save_model({k: z[k] for k in z.files if not k.startswith("_")}, None, 128, 1, 0, model)
        t0 = time.time()
        spec = assembly_spec(n, 16)
        made = write_assembly_dir(os.path.join(d, "img"), spec, 16, direct=True, processes=8,
                                  blocks=[blocks] * len(spec) if blocks > 1 else None)
        n = made["windows"]
        print("inputs (%d images, %d regions) written in %.1f s" % (n, made["regions"], time.time() - t0), flush=True)
        for mode in ["1"] * repeats + ["0"]:
            out = os.path.join(d, "out" + mode)
            shutil.rmtree(out, ignore_errors=True)
            t0 = time.time()
            r = subprocess.run([os.path.join(ROOT, "bin", "helen"), "polish", "-i", os.path.join(d, "img"), "-m", model, "-b", "256",
                                "-w", "8", "-t", threads, "-o", out, "-p", "asm", "-g"] + extra, cwd=ROOT,
                               env=dict(os.environ, HELEN_STITCH_PIPELINE=mode), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               text=True)
            dt = time.time() - t0
            info = [ln for ln in r.stderr.splitlines() if ln.startswith("INFO") and ("WINDOWS IN" in ln or "PIPELINED" in ln or "TIME" in ln or "WALL CLOCK" in ln or "COLLECTOR" in ln or "HOST PLAN" in ln)]
            print("HELEN_STITCH_PIPELINE=%s rc %d: polish wall %.2f s = %.0f windows/s (FASTA %d bytes)"
                  % (mode, r.returncode, dt, n / dt, os.path.getsize(os.path.join(out, "asm.fa")) if r.returncode == 0 else -1))
            print("\n".join("    " + ln for ln in info[-7:]))
            if r.returncode != 0:
                print(r.stderr[-3000:])
        a, b = (open(os.path.join(d, "out" + m, "asm.fa"), "rb").read() for m in "10")
        print("pipelined FASTA == two-phase FASTA:", a == b)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
