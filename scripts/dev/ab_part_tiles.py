"""Developer probe: where do the part-tile recurrences (HELEN_GRU_HALF8 / HELEN_GRU_QUARTER4) differ from gru_single8_kernel?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from helen_amd.engine import HelenEngine  # noqa: E402
from helen_amd.weights import make_weights  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "HELEN_GRU_QUARTER4"
eng = HelenEngine(make_weights(input_scale=1.0 / 64.0), device=0, max_windows=64)
torch.manual_seed(1)
for B, T in ((16, 1), (16, 2), (16, 3), (16, 100), (40, 7)):
    x = torch.rand((B, T, 90), device="cuda") * 255
    h = torch.rand((B, 2, 128), device="cuda") - 0.5
    out = {}
    for flag in ("0", "1"):
        os.environ["HELEN_GRU_HALF8"] = "0"
        os.environ["HELEN_GRU_QUARTER4"] = "0"
        os.environ["HELEN_GRU_PAIR"] = "0"
        os.environ[which] = flag
        out[flag] = [t.clone() for t in eng.chunk_forward(x, h)]
        torch.cuda.synchronize()
    for name, a, b in zip(("base", "rle", "hidden"), out["0"], out["1"]):
        d = (a - b).abs()
        bad = (d > 0).nonzero()
        print("B=%d T=%d %-6s max|d| %.3e  differing %d of %d" % (B, T, name, d.max().item(), bad.shape[0], d.numel()),
              "first:", bad[:6].tolist() if bad.shape[0] else "")
    if name == "hidden":
        d = (out["0"][2] - out["1"][2]).abs()
        print("   hidden: differing windows", sorted(set((d > 0).nonzero()[:, 0].tolist()))[:20],
              "dirs", sorted(set((d > 0).nonzero()[:, 1].tolist())),
              "units", sorted(set((d > 0).nonzero()[:, 2].tolist()))[:40])
