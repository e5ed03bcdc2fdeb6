#!/usr/bin/env python3
"""What bench.py profiles for `roofline.traffic`: helen_polish_batch calls of N windows on cuda:0 and nothing else.
    rocprofv3 --kernel-trace --pmc FETCH_SIZE ... -- python scripts/pmc_one_call.py fp32 4096 [calls=2]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helen_amd.engine import HelenEngine  # noqa: E402
from helen_amd.weights import make_weights  # noqa: E402

precision = sys.argv[1] if len(sys.argv) > 1 else "fp32"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 2
eng = HelenEngine(make_weights(input_scale=1.0 / 64.0), device=0, max_windows=n, precision=precision)
g = torch.Generator(device="cuda").manual_seed(20260928)
img = torch.randint(0, 256, (n, 1000, 90), dtype=torch.uint8, device="cuda", generator=g)
for _ in range(calls):
    eng.polish(img)
torch.cuda.synchronize()
eng.close()
