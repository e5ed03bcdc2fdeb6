"""How long the host takes to enqueue one helen_polish_batch call (77 launches) versus its device time."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from helen_amd.engine import HelenEngine  # noqa: E402
from helen_amd.weights import make_weights  # noqa: E402

for n in (256, 4096):
    eng = HelenEngine(make_weights(input_scale=1 / 64.0), device=0, max_windows=n)
    img = torch.randint(0, 256, (n, 1000, 90), dtype=torch.uint8, device="cuda")
    eng.polish(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.polish(img)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("n=%d: enqueue %.2f ms per call, device %.2f ms per call" % (n, (t1 - t0) * 100, (t2 - t0) * 100))
    eng.close()
