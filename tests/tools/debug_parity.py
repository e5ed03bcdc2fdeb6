import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from golden_cases import load_case
from helen_amd.engine import HelenEngine
from helen_amd.options import chunk_starts
case = sys.argv[1] if len(sys.argv) > 1 else "trace6"
w, img, g = load_case(case)
eng = HelenEngine(w, device=0, max_windows=64)
images = torch.from_numpy(img).cuda().float()
hidden = torch.zeros(img.shape[0], 2, 128, device="cuda")
for c, i in enumerate(chunk_starts()):
    base, rle, hidden = eng.chunk_forward(images[:, i:i + 100].contiguous(), hidden)
    eh = np.abs(hidden.cpu().numpy() - g["hidden"][c])
    msg = "chunk %2d hidden err max %.3g (fwd %.3g bwd %.3g)" % (c, eh.max(), eh[:, 0].max(), eh[:, 1].max())
    if c in (0, 9, 18):
        k = {0: 0, 9: 1, 18: 2}[c]
        eb = np.abs(base.cpu().numpy() - g["logit_base"][k])
        er = np.abs(rle.cpu().numpy() - g["logit_rle"][k])
        msg += "  logit err base %.3g rle %.3g; per-t max: %s" % (eb.max(), er.max(), np.round(er.max(axis=(0, 2))[:100:10], 5))
    print(msg)
b, r, ab, ar = eng.polish(torch.from_numpy(img).cuda(), want_acc=True)
ea = np.abs(ab.cpu().numpy()[:3] - g["acc_base"])
print("acc err per 50-block:", np.round(ea.reshape(3, 20, 50, 5).max(axis=(0, 2, 3)), 6))
print("labels mismatches base", int((b.cpu().numpy() != g["bases"]).sum()), "rle", int((r.cpu().numpy() != g["rles"]).sum()))
