"""helen_polish_host from PAGEABLE caller memory under the three locking rules ($HELEN_HOST_LOCK = none | own | all, read at
model creation), and from page-locked memory: windows/s of one call over 8 sub-batches of 4096 windows.
   python scripts/dev/host_lock_modes.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from helen_amd.engine import HelenEngine  # noqa: E402
from helen_amd.weights import make_weights  # noqa: E402

n = 8 * 4096
img = np.random.default_rng(0).integers(0, 256, (n, 1000, 90), dtype=np.uint8)
w = make_weights(input_scale=1 / 64.)
ref = None
for mode in ("none", "own", "all", "pinned"):
    os.environ["HELEN_HOST_LOCK"] = "own" if mode == "pinned" else mode
    eng = HelenEngine(w, device=0, max_windows=4096)
    if mode == "pinned":
        src = torch.from_numpy(img).pin_memory().numpy()
        out = (torch.empty((n, 1000), dtype=torch.uint8).pin_memory().numpy(), torch.empty((n, 1000), dtype=torch.uint8).pin_memory().numpy())
    else:
        src, out = img, None
    eng.polish_host(src[:4096])
    rates = []
    for rep in range(3):
        t0 = time.time()
        hb, hr = eng.polish_host(src, out=out)
        rates.append(n / (time.time() - t0))
    if ref is None:
        ref = (hb.copy(), hr.copy())
    same = np.array_equal(hb, ref[0]) and np.array_equal(hr, ref[1])
    print("%-6s: %s windows/s  (labels equal: %s)" % (mode, ", ".join("%.0f" % r for r in rates), same), flush=True)
    eng.close()
