import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle
from golden_cases import load_case
from helen_amd.engine import HelenEngine
from helen_amd.options import chunk_starts
for case in ("trace6", "small_input6"):
    w, img, g = load_case(case)
    eng = HelenEngine(w, device=0, max_windows=64, precision="bf16")
    images = torch.from_numpy(img).cuda()
    xf = images.float()
    oracle.set_precision("bf16"); emu = oracle.polish_batch(w, img, traces=True); oracle.set_precision("fp32")
    hidden = torch.zeros(img.shape[0], 2, 128, device="cuda")
    for c, i in enumerate(chunk_starts()):
        base, rle, hidden = eng.chunk_forward(xf[:, i:i + 100].contiguous(), hidden)
        if c in (0, 1, 9, 18):
            print(case, "chunk", c, "vs emu: hidden %.3g base %.3g rle %.3g" % (
                np.abs(hidden.cpu().numpy() - emu["hidden"][c]).max(),
                np.abs(base.cpu().numpy() - emu["logit_base"][c]).max(),
                np.abs(rle.cpu().numpy() - emu["logit_rle"][c]).max()),
                "| frac > 1e-2: %.4f" % (np.abs(rle.cpu().numpy() - emu["logit_rle"][c]) > 1e-2).mean())
    b, r = eng.polish(images)
    print(case, "label mismatch vs emu", (b.cpu().numpy() != emu["bases"]).mean(), (r.cpu().numpy() != emu["rles"]).mean(),
          "vs fp32 ref", (b.cpu().numpy() != g["bases"]).mean(), (r.cpu().numpy() != g["rles"]).mean())
