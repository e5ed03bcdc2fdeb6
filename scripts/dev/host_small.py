"""Developer soak: helen_polish_host on small pageable arrays from the Python heap (label rows of a few KiB share their pages
with other heap objects), many times, with Python allocating and freeing around the calls.
   python scripts/dev/host_small.py [iterations]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from helen_amd.engine import HelenEngine  # noqa: E402
from helen_amd.weights import make_weights  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(3)
eng = HelenEngine(make_weights(input_scale=1.0 / 64.0), device=0, max_windows=600)
img = rng.integers(0, 256, (600, 1000, 90), dtype=np.uint8)
dev = torch.from_numpy(img).cuda()
want_b, want_r = [t.cpu().numpy() for t in eng.polish(dev)]
junk = []
bad = 0
for i in range(iters):
    n = int(rng.integers(1, 600)) if i % 3 else int(rng.integers(1, 40))
    junk.append(bytearray(int(rng.integers(100, 50000))))          # heap churn around the label arrays
    if len(junk) > 20:
        del junk[:int(rng.integers(1, 15))]
    hb, hr = eng.polish_host(img[:n])
    if not (np.array_equal(hb, want_b[:n]) and np.array_equal(hr, want_r[:n])):
        bad += 1
    if i % 50 == 49:
        print("%d calls, %d wrong" % (i + 1, bad), flush=True)
eng.close()
print("done: %d calls, %d wrong" % (iters, bad))
sys.exit(1 if bad else 0)
