#!/usr/bin/env python3
"""Developer probe: throughput and per-kernel-class time of helen_polish_batch at several sizes."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helen_amd.engine import HelenEngine  # noqa: E402
from helen_amd.weights import make_weights  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--windows", type=int, nargs="+", default=[256, 2048, 4096])
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--precision", default="fp32")
args = ap.parse_args()

w = make_weights(input_scale=1.0 / 64.0)
for n in args.windows:
    eng = HelenEngine(w, device=0, max_windows=n, precision=args.precision)
    g = torch.Generator(device="cuda").manual_seed(1)
    img = torch.randint(0, 256, (n, 1000, 90), dtype=torch.uint8, device="cuda", generator=g)
    eng.polish(img)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.iters):
        eng.polish(img)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / args.iters
    eng.set_profiling(["pack", "gemm_enc", "gru_enc", "gemm_dec", "gru_dec", "heads"])
    eng.reset_kernel_stats()
    eng.polish(img)
    torch.cuda.synchronize()
    st = eng.kernel_stats()
    eng.set_profiling([])
    wps = n / dt
    print("n=%d  %.1f ms/call  %.0f windows/s  path_frac=%.3f  dev_bytes=%.2f GB" %
          (n, dt * 1e3, wps, wps * 1.7724416e9 / 157.3e12, eng.device_bytes / 1e9))
    for k, (ms, cnt) in st.items():
        print("   %-9s %8.3f ms total  %4d launches  %8.3f ms avg" % (k, ms, cnt, ms / max(cnt, 1)))
    eng.close()
    del img
    torch.cuda.empty_cache()
