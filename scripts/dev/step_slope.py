"""Developer probe: recurrence launch time versus chunk length T (operator entry, 4096 windows): the slope is the
cost of a step, the intercept the fixed cost of a launch (prologue, epilogue, launch gap)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from helen_amd.engine import HelenEngine  # noqa: E402
from helen_amd.weights import make_weights  # noqa: E402

n = 4096
eng = HelenEngine(make_weights(input_scale=1 / 64.0), device=0, max_windows=n)
h = torch.zeros(n, 2, 128, device="cuda")
res = {}
for T in (10, 25, 50, 100):
    x = torch.rand(n, T, 90, device="cuda") * 255
    for _ in range(2):
        eng.chunk_forward(x, h)
    torch.cuda.synchronize()
    eng.set_profiling(["gru_enc", "gemm_dec", "gru_dec", "gemm_enc"])
    eng.reset_kernel_stats()
    for _ in range(5):
        eng.chunk_forward(x, h)
    torch.cuda.synchronize()
    st = eng.kernel_stats()
    eng.set_profiling([])
    res[T] = {k: st[k][0] / st[k][1] for k in ("gru_enc", "gru_dec", "gemm_dec", "gemm_enc")}
    print("T=%3d  " % T + "  ".join("%s %.4f ms" % kv for kv in res[T].items()))
for k in ("gru_enc", "gru_dec", "gemm_dec", "gemm_enc"):
    slope = (res[100][k] - res[50][k]) / 50
    print("%-9s per step %.3f us, fixed %.1f us" % (k, slope * 1e3, (res[100][k] - 100 * slope) * 1e3))
