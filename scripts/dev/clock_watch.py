#!/usr/bin/env python3
"""Developer probe (GPU box): shader clock and socket power while each precision mode's kernels run back to back for a few
seconds -- rocm-smi sampled from a side thread -- beside the idle reading.  Evidence for profiles/r06_region_anatomy.txt 4
(the bf16-pipe kernels run at a lower clock in a stream of launches than alone).
    python scripts/dev/clock_watch.py [seconds=6]"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from helen_amd.engine import HelenEngine  # noqa: E402
from helen_amd.weights import make_weights  # noqa: E402


def sample():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out[out.index("{"):])
        card = d[sorted(d)[0]]
        sclk = next((v for k, v in card.items() if k.startswith("sclk")), "?")
        power = next((v for k, v in card.items() if "ower" in k and "W" in k), "?")
        return str(sclk), str(power)
    except Exception as e:       # noqa: BLE001 - a probe: say what went wrong and go on
        return "?", repr(e)[:80]


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    print("idle:", sample())
    w = make_weights(input_scale=1.0 / 64.0)
    for precision, n in (("fp32", 4096), ("fp32x3", 4096), ("bf16", 8192)):
        eng = HelenEngine(w, device=0, max_windows=n, precision=precision)
        img = torch.randint(0, 256, (n, 1000, 90), dtype=torch.uint8, device="cuda")
        eng.polish(img)
        torch.cuda.synchronize()
        seen, stop = [], threading.Event()

        def watch():
            while not stop.is_set():
                seen.append(sample())
                time.sleep(0.05)

        t = threading.Thread(target=watch)
        t.start()
        t0, calls = time.time(), 0
        while time.time() - t0 < seconds:
            for _ in range(8):
                eng.polish(img)
            torch.cuda.synchronize()
            calls += 8
        dt = time.time() - t0
        stop.set()
        t.join()
        print("%s: %.0f windows/s over %.1f s; rocm-smi samples (sclk, power): %s" % (precision, calls * n / dt, dt, seen[1:-1][:12]))
        del eng
    print("idle again:", sample())


if __name__ == "__main__":
    main()
