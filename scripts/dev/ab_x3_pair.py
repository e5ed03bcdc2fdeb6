#!/usr/bin/env python3
"""Developer probe (GPU box): the fp32x3 recurrence with one tile per workgroup (gru_x3_kernel, HELEN_X3_PAIR=0) against two
tiles per workgroup with the gate math inside the other tile's MFMA stream (gru_x3_il_kernel, =1): the outputs must be
EQUAL (labels, accumulated softmax, operator-entry logits and hidden state at T = 1, 2, 3, 37, 100, an odd tile count),
and the launch times side by side.
    python scripts/dev/ab_x3_pair.py [windows=4096]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from helen_amd.engine import HelenEngine  # noqa: E402
from helen_amd.weights import make_weights  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    w = make_weights(input_scale=1.0 / 64.0)
    g = torch.Generator(device="cuda").manual_seed(3)
    img = torch.randint(0, 256, (n, 1000, 90), dtype=torch.uint8, device="cuda", generator=g)
    eng = HelenEngine(w, device=0, max_windows=n, precision="fp32x3")
    got, times = {}, {}
    for pair in ("0", "1"):
        os.environ["HELEN_X3_PAIR"] = pair
        eng.reload_overrides()
        out = [eng.polish(img, want_acc=True), eng.polish(img[:n - 16 - 3], want_acc=True)]
        for T in (1, 2, 3, 37, 100):
            x = torch.rand((min(n, 2048), T, 90), device="cuda", generator=g) * 255 if False else torch.rand((min(n, 2048), T, 90), device="cuda") * 0 + \
                torch.arange(T, device="cuda", dtype=torch.float32)[None, :, None] * 0.37 + torch.arange(90, device="cuda", dtype=torch.float32)[None, None, :] * 0.11
            x = x + torch.arange(min(n, 2048), device="cuda", dtype=torch.float32)[:, None, None] * 0.013
            h = torch.sin(torch.arange(min(n, 2048) * 256, device="cuda", dtype=torch.float32)).reshape(-1, 2, 128) * 0.5
            out.append(eng.chunk_forward(x, h))
        torch.cuda.synchronize()
        got[pair] = out
        eng.polish(img)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            eng.polish(img)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 5
        eng.set_profiling(["pack", "gemm_enc", "gru_enc", "gemm_dec", "gru_dec", "heads"])
        eng.reset_kernel_stats()
        eng.polish(img)
        torch.cuda.synchronize()
        st = eng.kernel_stats()
        eng.set_profiling([])
        times[pair] = (dt, st)
        print("HELEN_X3_PAIR=%s: %.2f ms per call = %.0f windows/s; gru_enc %.4f ms, gru_dec %.4f ms, gemm_dec %.4f ms per launch"
              % (pair, dt * 1e3, n / dt, st["gru_enc"][0] / st["gru_enc"][1], st["gru_dec"][0] / st["gru_dec"][1],
                 st["gemm_dec"][0] / st["gemm_dec"][1]))
    bad = 0
    for k, (a, b) in enumerate(zip(got["0"], got["1"])):
        for u, v_ in zip(a, b):
            if not torch.equal(u, v_):
                bad += 1
                d = (u.float() - v_.float()).abs()
                print("output group %d differs: max |diff| %.3g, %d of %d elements" % (k, float(d.max()), int((d > 0).sum()), d.numel()))
    print("two tiles per workgroup == one tile per workgroup:", bad == 0)
    eng.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
