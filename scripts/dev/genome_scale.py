#!/usr/bin/env python3
"""Rehearsal of BASELINE.json configs[2] on the hardware there is (GPU box, ONE MI355X): `bin/helen polish -g` with 8 ranks
sharing the device over a simulated assembly of whole-genome size -- 3.0 M windows / 1 M regions, images chunked and
deflated (what keeps 270 GB of pixels inside a RAM-backed directory) -- in steps over the first 20 %, 50 % and all of
the image files.  Per step: the command's wall clock and its own stage report, every rank's stage seconds, the stitch
collectors' report, peak resident memory of parent / ranks / collectors (the command prints VmHWM), the high-water marks
of the memory cgroup and of /dev/shm sampled beside it, oom kills before / after, and the FASTA's sha1 against `helen
stitch` run afterwards on the finished prediction files.
    python scripts/dev/genome_scale.py [windows=3000000] [ranks=8] [steps=0.2,0.5,1.0] [gzip=1] [threads=16]
Writes gpurun_out/genome_scale.json and prints the same as text."""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from helen_amd.model_handler import ModelHandler  # noqa: E402
from helen_amd.synthetic import assembly_spec, write_assembly_dir  # noqa: E402

CG = "/sys/fs/cgroup"


def _int(path):
    try:
        return int(open(path).read().split()[0])
    except (OSError, ValueError, IndexError):
        return None


def oom_kills():
    try:
        for ln in open(os.path.join(CG, "memory.events")):
            if ln.startswith("oom_kill "):
                return int(ln.split()[1])
    except OSError:
        pass
    return None


class Watch(threading.Thread):
    """High-water marks of the memory cgroup and of /dev/shm while a command runs."""

    def __init__(self):
        threading.Thread.__init__(self, daemon=True)
        self.stop = threading.Event()
        self.peak_cgroup = self.peak_shm = 0

    def run(self):
        while not self.stop.is_set():
            cur = _int(os.path.join(CG, "memory.current")) or 0
            st = os.statvfs("/dev/shm")
            used = (st.f_blocks - st.f_bfree) * st.f_frsize
            self.peak_cgroup, self.peak_shm = max(self.peak_cgroup, cur), max(self.peak_shm, used)
            self.stop.wait(0.5)


def sha1(path):
    h = hashlib.sha1()
    with open(path, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                return h.hexdigest()
            h.update(b)


def main():
    windows = int(sys.argv[1]) if len(sys.argv) > 1 else 3000000
    ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    steps = [float(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0.2,0.5,1.0").split(",")]
    gz = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    threads = sys.argv[5] if len(sys.argv) > 5 else "16"
    report = {"what": "bin/helen polish -g -d_ids %s -t %s -w 8 -b 256 on a simulated assembly, images deflated (gzip %d), "
                      "all ranks on ONE MI355X" % (",".join("0" * ranks), threads, gz), "steps": []}
    d = tempfile.mkdtemp(prefix="helen_genome_", dir="/dev/shm")
    try:
        model = os.path.join(d, "model.pkl")
        z = np.load(os.path.join(ROOT, "tests", "golden", "trained_synth.npz"))
        ModelHandler.save_model({k: z[k] for k in z.files if not k.startswith("_")}, None, 128, 1, 0, model)
        n_files = 16 * ranks
        t0 = time.time()
        spec = assembly_spec(windows, n_files)
        made = write_assembly_dir(os.path.join(d, "img"), spec, n_files, direct=gz == 0, gzip=gz or None,
                                  processes=min(16, os.cpu_count() or 1))
        files = sorted(made["files"])
        size = sum(os.path.getsize(f) for f in files)
        report["inputs"] = {"windows": made["windows"], "regions": made["regions"], "contigs": len(spec), "files": len(files),
                            "bytes": size, "bytes_per_window": round(size / float(made["windows"]), 1),
                            "seconds_to_write": round(time.time() - t0, 1), "memory_cgroup_limit": _int(os.path.join(CG, "memory.max"))}
        print("inputs: %s" % json.dumps(report["inputs"]), flush=True)
        per_file = made["windows_per_file"]
        for frac in steps:
            k = max(ranks, int(round(frac * len(files))))
            sub = os.path.join(d, "img_%d" % k)
            os.makedirs(sub)
            for f in files[:k]:
                os.symlink(f, os.path.join(sub, os.path.basename(f)))
            n = sum(per_file[:k])
            out = os.path.join(d, "out_%d" % k)
            cmd = [sys.executable, os.path.join(ROOT, "bin", "helen"), "polish", "-i", sub, "-m", model, "-b", "256", "-w", "8", "-t", threads,
                   "-o", out, "-p", "asm"]
            if os.environ.get("GENOME_SCALE_HOST_PATH") == "1":      # plumbing check on a machine without a GPU (tiny sizes)
                cmd += ["-c", str(ranks)]
            else:
                cmd += ["-g"] + (["-d_ids", ",".join("0" * ranks)] if ranks > 1 else [])
            oom0 = oom_kills()
            w = Watch()
            w.start()
            t0 = time.time()
            r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            dt = time.time() - t0
            w.stop.set()
            w.join()
            info = [ln[6:] for ln in r.stderr.splitlines() if ln.startswith("INFO: ") and any(
                k_ in ln for k_ in ("WALL CLOCK", "WINDOWS IN", "RANK ", "PEAK RESIDENT", "COLLECTOR", "HOST PLAN", "JOIN", "HOST-BOUND", "PLAN:"))]
            warn = [ln for ln in r.stderr.splitlines() if ln.startswith(("WARNING", "ERROR"))]
            step = {"files": k, "windows": n, "returncode": r.returncode, "seconds": round(dt, 2), "windows_per_s": round(n / dt, 1),
                    "peak_memory_cgroup_GB": round(w.peak_cgroup / 1e9, 1), "peak_dev_shm_GB": round(w.peak_shm / 1e9, 1),
                    "oom_kill_before_after": [oom0, oom_kills()], "report": info, "warnings": warn[:10]}
            fasta = os.path.join(out, "asm.fa")
            if r.returncode == 0 and os.path.isfile(fasta):
                step["fasta_bytes"] = os.path.getsize(fasta)
                step["fasta_sha1"] = sha1(fasta)
                pred = [os.path.join(out, p) for p in os.listdir(out) if p.startswith("predictions_")][0]
                step["prediction_bytes"] = sum(os.path.getsize(os.path.join(pred, f)) for f in os.listdir(pred))
                t0 = time.time()
                r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "helen"), "stitch", "-i", pred, "-o", os.path.join(d, "two_phase_%d" % k),
                                     "-p", "asm", "-t", threads], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                step["helen_stitch_seconds"] = round(time.time() - t0, 2)
                f2 = os.path.join(d, "two_phase_%d" % k, "asm.fa")
                step["fasta_equals_helen_stitch"] = r2.returncode == 0 and os.path.isfile(f2) and sha1(f2) == step["fasta_sha1"]
                if r2.returncode:
                    step["helen_stitch_error"] = r2.stderr[-800:]
                shutil.rmtree(os.path.join(d, "two_phase_%d" % k), ignore_errors=True)
            else:
                step["stderr_tail"] = r.stderr[-3000:]
            shutil.rmtree(out, ignore_errors=True)
            report["steps"].append(step)
            print(json.dumps(step, indent=1), flush=True)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            json.dump(report, open(os.path.join(ROOT, "gpurun_out", "genome_scale.json"), "w"), indent=1)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
