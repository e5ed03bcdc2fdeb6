#!/usr/bin/env python3
"""Developer probe: error of each arithmetic mode against the reference goldens AND against a
float64 evaluation of the same network (who is closer to the exact answer?)."""
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from golden_cases import load_case  # noqa: E402
from helen_amd.engine import HelenEngine  # noqa: E402
from helen_amd.options import chunk_starts  # noqa: E402


def f64_reference(w, img):
    """The path in float64 on the CPU with torch (same equations as oracle/helen_oracle.c)."""
    W = {k: torch.from_numpy(v).double() for k, v in w.items()}
    x = torch.from_numpy(img).double()
    B = x.shape[0]
    hid = torch.zeros(B, 2, 128, dtype=torch.float64)
    out_h, out_l = [], []

    def gru(xs, h, pre, rev):
        wi, wh, bi, bh = (W[pre + ".weight_ih_l0" + rev], W[pre + ".weight_hh_l0" + rev],
                          W[pre + ".bias_ih_l0" + rev], W[pre + ".bias_hh_l0" + rev])
        T = xs.shape[1]
        ys = [None] * T
        for s in range(T):
            t = T - 1 - s if rev else s
            gi = xs[:, t] @ wi.T + bi
            gh = h @ wh.T + bh
            r = torch.sigmoid(gi[:, :128] + gh[:, :128])
            z = torch.sigmoid(gi[:, 128:256] + gh[:, 128:256])
            n = torch.tanh(gi[:, 256:] + r * gh[:, 256:])
            h = (1 - z) * n + z * h
            ys[t] = h
        return torch.stack(ys, 1), h
    for i in chunk_starts():
        xc = x[:, i:i + 100]
        yf, hf = gru(xc, hid[:, 0], "gru_encoder", "")
        yb, hb = gru(xc, hid[:, 1], "gru_encoder", "_reverse")
        y1 = torch.cat([yf, yb], 2)
        yf, hf = gru(y1, hf, "gru_decoder", "")
        yb, hb = gru(y1, hb, "gru_decoder", "_reverse")
        y2 = torch.cat([yf, yb], 2)
        hid = torch.stack([hf, hb], 1)
        out_h.append(hid.clone())
        out_l.append(torch.cat([y2 @ W["dense1_base.weight"].T + W["dense1_base.bias"],
                                y2 @ W["dense2_rle.weight"].T + W["dense2_rle.bias"]], 2))
    return torch.stack(out_h).numpy(), torch.stack(out_l).numpy()


for case in ("trace6", "small_input6"):
    w, img, g = load_case(case)
    h64, l64 = f64_reference(w, img)
    print(case, "reference goldens (torch fp32 CPU) vs float64: hidden %.3g logits %.3g"
          % (np.abs(g["hidden"] - h64).max(),
             max(np.abs(np.concatenate([g["logit_base"][k], g["logit_rle"][k]], 2) - l64[c]).max()
                 for k, c in enumerate((0, 9, 18)))))
    for prec in ("fp32", "fp32x3", "bf16"):
        eng = HelenEngine(w, device=0, max_windows=64, precision=prec)
        xf = torch.from_numpy(img).cuda().float()
        hidden = torch.zeros(img.shape[0], 2, 128, device="cuda")
        eh = el = 0.0
        for c, i in enumerate(chunk_starts()):
            base, rle, hidden = eng.chunk_forward(xf[:, i:i + 100].contiguous(), hidden)
            eh = max(eh, np.abs(hidden.cpu().numpy() - h64[c]).max())
            el = max(el, np.abs(torch.cat([base, rle], 2).cpu().numpy() - l64[c]).max())
        b, r = eng.polish(torch.from_numpy(img).cuda())
        mis = int((b.cpu().numpy() != g["bases"]).sum() + (r.cpu().numpy() != g["rles"]).sum())
        print("   %-7s vs float64: hidden %.3g logits %.3g   label mismatches vs reference: %d" % (prec, eh, el, mis))
        eng.close()
