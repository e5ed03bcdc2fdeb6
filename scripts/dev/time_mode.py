#!/usr/bin/env python3
"""Developer probe (GPU box): one precision mode's call time and recurrence launch times, for the library HELEN_HIP_LIB
names (a timing build) or the tree's -- alternate the two on one box for an A/B (profiles/r06_bf16_head_out_ab.txt).
    [HELEN_HIP_LIB=$PWD/build/lib_<variant>.so] python scripts/dev/time_mode.py fp32|fp32x3|bf16 <windows per call>"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from helen_amd.engine import HelenEngine
from helen_amd.weights import make_weights
prec, n = sys.argv[1], int(sys.argv[2])
eng = HelenEngine(make_weights(input_scale=1/64.), device=0, max_windows=n, precision=prec)
img = torch.randint(0, 256, (n, 1000, 90), dtype=torch.uint8, device="cuda")
for _ in range(3): eng.polish(img)
torch.cuda.synchronize()
t0=time.time()
for _ in range(8): eng.polish(img)
torch.cuda.synchronize(); dt=(time.time()-t0)/8
eng.set_profiling(["gru_enc","gru_dec"]); eng.reset_kernel_stats(); eng.polish(img); torch.cuda.synchronize(); st=eng.kernel_stats()
print("%s %s: %.2f ms per call = %.0f windows/s; gru_enc %.4f ms, gru_dec %.4f ms" % (os.environ.get("HELEN_HIP_LIB","tree"), prec, dt*1e3, n/dt, st["gru_enc"][0]/st["gru_enc"][1], st["gru_dec"][0]/st["gru_dec"][1]))
