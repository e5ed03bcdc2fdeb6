"""Developer probe: polish the same seeded windows and either save (labels + accumulators) or compare with a
saved run bit for bit -- for A/B runs of kernel variants selected by environment / HELEN_HIP_LIB.
    python scripts/dev/ab_equal.py save /tmp/a.pt [n] ; HELEN_GRU_PAIR=1 python scripts/dev/ab_equal.py cmp /tmp/a.pt [n]
HELEN_AB_PRECISION=bf16 / fp32x3 selects the engine's precision mode."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from helen_amd.engine import HelenEngine  # noqa: E402
from helen_amd.weights import make_weights  # noqa: E402

mode, path = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
cap = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
g = torch.Generator(device="cuda").manual_seed(11)
img = torch.randint(0, 256, (n, 1000, 90), dtype=torch.uint8, device="cuda", generator=g)
eng = HelenEngine(make_weights(seed=20260928, head_scale=8.0, input_scale=1 / 64.0), device=0, max_windows=cap,
                  precision=os.environ.get("HELEN_AB_PRECISION", "fp32"))
out = [t.cpu() for t in eng.polish(img, want_acc=True)]
x = torch.rand((min(n, 64), 100, 90), device="cuda", generator=g)
h = torch.rand((min(n, 64), 2, 128), device="cuda", generator=g) - 0.5
out += [t.cpu() for t in eng.chunk_forward(x, h)]
torch.cuda.synchronize()
if mode == "save":
    torch.save(out, path)
    print("saved", path)
else:
    ref = torch.load(path)
    names = ["bases", "rles", "acc_base", "acc_rle", "op_base", "op_rle", "op_hidden"]
    bad = 0
    for nm, a, b in zip(names, ref, out):
        eq = torch.equal(a, b)
        d = (a.float() - b.float()).abs().max().item()
        print("%-9s %s  max|diff| %.3g" % (nm, "EQUAL" if eq else "DIFFERENT", d))
        bad += not eq
    sys.exit(1 if bad else 0)
