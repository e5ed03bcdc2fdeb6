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

# the queueing entry: the same windows handed over as loader batches of 256 (helen_polish_submit / helen_polish_flush)
os.environ.pop("HELEN_HOST_LOCK", None)
eng = HelenEngine(w, device=0, max_windows=4096)
ob, orr = np.empty((n, 1000), np.uint8), np.empty((n, 1000), np.uint8)
for lo in range(0, 4096, 256):
    eng.submit(img[lo:lo + 256], (ob[lo:lo + 256], orr[lo:lo + 256]))
eng.flush()
rates = []
for rep in range(3):
    t0 = time.time()
    for lo in range(0, n, 256):
        eng.submit(img[lo:lo + 256], (ob[lo:lo + 256], orr[lo:lo + 256]))
    eng.flush()
    rates.append(n / (time.time() - t0))
print("queued: %s windows/s  (128 submissions of 256 windows + flush; labels equal: %s)"
      % (", ".join("%.0f" % r for r in rates), np.array_equal(ob, ref[0]) and np.array_equal(orr, ref[1])), flush=True)
t0 = time.time()
for lo in range(0, 4096, 256):
    eng.polish_host(img[lo:lo + 256])
print("for comparison, helen_polish_host per batch of 256: %.0f windows/s" % (4096 / (time.time() - t0)))
eng.close()
