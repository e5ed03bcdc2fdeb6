#!/usr/bin/env python3
"""SURVEY.md 8f-2 / BASELINE.json configs[4], one command away.  Real MarginPolish images are not available offline; with a
directory of them and a model:

    python scripts/real_images_check.py <image_dir> <model.pkl> [windows=2000] [--no-gpu]

  1. `check_images --strict`: must say "ready" (README.md lists what in its report would FALSIFY the schema inferred from
     the reference's reader, dataloader_predict.py:64-70);
  2. `helen polish -g` on the directory: the FASTA of the pipelined command must equal `helen stitch` on the prediction files
     it wrote;
  3. on the first `windows` windows every label of the MI355X path must equal the host path's (libhelen_cpu.so: an independent
     implementation of the same arithmetic, itself pinned by the reference's goldens) -- up to sub-fp32-resolution ties.
--no-gpu runs 1 and 2 on the host path only (plumbing check on a machine without the device).
The reference's own chain on the same files (tests/golden/make_golden_polish.py is the recipe) needs its environment and is
the last step.  Exit code 0 = all three hold."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def helen(args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "helen")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if p.returncode != 0:
        sys.stderr.write(p.stderr[-3000:])
        sys.exit("helen %s failed" % args[0])
    return p


def main():
    args = [a for a in sys.argv[1:] if a != "--no-gpu"]
    gpu = "--no-gpu" not in sys.argv
    if len(args) < 2:
        sys.exit(__doc__)
    images, model = args[0], args[1]
    n = int(args[2]) if len(args) > 2 else 2000
    from helen_amd.check_images import image_directory_report
    report = image_directory_report(images, strict=True)
    print("1. check_images --strict: %s; %d files, %d images, %d read through the product reader; storage %s"
          % (report["verdict"], len(report["files"]), report["images"], report["images_read"],
             sorted({str(f.get("storage_class")) for f in report["files"]})))
    ok = report["refusals"] == 0 and report["problems"] == 0
    with tempfile.TemporaryDirectory(prefix="helen_real_") as d:
        out = os.path.join(d, "out")
        r = helen(["polish", "-i", images, "-m", model, "-b", "256", "-w", "8", "-t", "16", "-o", out, "-p", "real"] + (["-g"] if gpu else []))
        pred = [os.path.join(out, x) for x in os.listdir(out) if x.startswith("predictions_")][0]
        helen(["stitch", "-i", pred, "-o", os.path.join(d, "two_phase"), "-p", "real", "-t", "16"])
        a = open(os.path.join(out, "real.fa"), "rb").read()
        b = open(os.path.join(d, "two_phase", "real.fa"), "rb").read()
        same = a == b and len(a) > 0
        print("2. polish%s: FASTA %d bytes; equals `helen stitch` on its prediction files: %s" % (" -g" if gpu else "", len(a), same))
        for ln in r.stderr.splitlines():
            if "WINDOWS IN" in ln or "WALL CLOCK" in ln or "SHORTCUT" in ln:
                print("   " + ln[:240])
        ok = ok and same
    if gpu:
        import torch

        from helen_amd.checkpoint import load_simple_model_state
        from helen_amd.cpu_engine import CpuEngine
        from helen_amd.engine import HelenEngine
        from helen_amd.sequence_dataset import SequenceDataset
        data = SequenceDataset(images)
        n = min(n, len(data))
        batch = np.stack([np.asarray(data[i][4]) for i in range(n)])
        state = load_simple_model_state(model)[0]
        eng = HelenEngine(state, device=0, max_windows=min(4096, max(16, n)))
        gb, gr = eng.polish(torch.from_numpy(batch).cuda())
        torch.cuda.synchronize()
        eng.close()
        hb, hr = CpuEngine(state, threads=16).polish_host(batch)
        diff = int((gb.cpu().numpy() != hb).sum() + (gr.cpu().numpy() != hr).sum())
        print("3. MI355X path against the host path on %d real windows: %d of %d labels differ" % (n, diff, 2 * n * 1000))
        ok = ok and diff <= max(1, int(1e-6 * 2 * n * 1000))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
