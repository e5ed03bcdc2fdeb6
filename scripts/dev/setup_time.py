"""Developer probe (GPU box): where the MODEL + ENGINE SET-UP time of predict() goes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
t0 = time.time()
import torch  # noqa: E402
t1 = time.time()
from helen_amd.engine import HelenEngine  # noqa: E402
from helen_amd.weights import make_weights  # noqa: E402
t2 = time.time()
torch.cuda.set_device(0)
torch.zeros(1, device="cuda")
torch.cuda.synchronize()
t3 = time.time()
w = make_weights()
t4 = time.time()
e = HelenEngine(w, device=0, max_windows=4096)
torch.cuda.synchronize()
t5 = time.time()
img = torch.zeros((4096, 1000, 90), dtype=torch.uint8, device="cuda")
e.polish(img)
torch.cuda.synchronize()
t6 = time.time()
e.polish(img)
torch.cuda.synchronize()
t7 = time.time()
print("import torch %.2f  import helen_amd %.2f  context %.2f  weights %.2f  engine create %.2f  first call %.3f  "
      "second call %.3f" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6))
