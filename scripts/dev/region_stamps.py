#!/usr/bin/env python3
"""Developer probe (GPU box): where a region of the two-tile kernels (gru_x3_il_kernel, gru_fused_bf16_il_kernel) goes, per
wave of workgroup 0.  Needs the stamped library scripts/dev/make_stamp_probe.py builds (build/lib_stamps.so).
    HELEN_HIP_LIB=$PWD/build/lib_stamps.so python scripts/dev/region_stamps.py [windows=4096] [fp32x3|bf16|fp32]"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from helen_amd.engine import HelenEngine  # noqa: E402
from helen_amd.weights import make_weights  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    precision = sys.argv[2] if len(sys.argv) > 2 else "fp32x3"
    lib = ctypes.CDLL(os.environ["HELEN_HIP_LIB"])
    eng = HelenEngine(make_weights(input_scale=1.0 / 64.0), device=0, max_windows=n, precision=precision)
    img = torch.randint(0, 256, (n, 1000, 90), dtype=torch.uint8, device="cuda")
    eng.polish(img)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        eng.polish(img)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 5
    eng.set_profiling(["gru_enc", "gru_dec"])
    eng.reset_kernel_stats()
    eng.polish(img)
    torch.cuda.synchronize()
    st = eng.kernel_stats()
    eng.set_profiling([])
    print(precision + " %d windows: %.2f ms per call = %.0f windows/s; gru_enc %.4f ms, gru_dec %.4f ms per launch"
          % (n, dt * 1e3, n / dt, st["gru_enc"][0] / st["gru_enc"][1], st["gru_dec"][0] / st["gru_dec"][1]))
    buf = (ctypes.c_ulonglong * (2 * 2 * 8 * 16))()
    if precision == "fp32":
        # gru_pair_kernel: a half-step = M phase of one tile (96 fp32 MFMAs per wave) | barrier | that tile's gate math.
        # [0] half-step start, [1] after the M phase, [2] in front of the barrier, [3] behind it; sf[m] in front of K16 group m
        lib.helen_debug_pair(buf)
        a = torch.tensor(list(buf), dtype=torch.int64).reshape(2, 2, 8, 4, 4)
        lib.helen_debug_pairf(buf)
        f = torch.tensor(list(buf), dtype=torch.int64).reshape(2, 2, 8, 16)
        for dec in (0, 1):
            s = a[dec, 0]
            base = int(s[:, 0, 0].min())
            print("%s dir 0, workgroup 0, last steady trip: per half-step  start (relative)  M phase  sums + LDS drain  wait at the barrier  gates (to the next start)"
                  % ("decoder" if dec else "encoder"))
            for v in range(8):
                parts = []
                for r in range(4):
                    t = [int(z) for z in s[v, r]]
                    nxt = int(s[v, r + 1, 0]) if r < 3 else None
                    parts.append("%6d %5d %4d %5d %5s" % (t[0] - base, t[1] - t[0], t[2] - t[1], t[3] - t[2], "-" if nxt is None else str(nxt - t[3])))
                print("  %d    | %s" % (v, " | ".join(parts)))
            print("%s dir 0, half-step r0: start -> group 0 | cycles per K16 group (12 MFMAs each) | last group -> M end" % ("decoder" if dec else "encoder"))
            for v in range(8):
                t = [int(z) for z in f[dec, 0, v]]
                print("  %d    | %4d | %s | %4d" % (v, t[0] - int(s[v, 0, 0]), " ".join("%4d" % (t[k + 1] - t[k]) for k in range(7)), int(s[v, 0, 1]) - t[7]))
        return
    rc = (lib.helen_debug_x3 if precision == "fp32x3" else lib.helen_debug_bf16)(buf)
    print(precision, "stamps rc", rc)
    a = torch.tensor(list(buf), dtype=torch.int64).reshape(2, 2, 8, 4, 4)      # [dec][dir][wave][region][stamp]
    for dec in (0, 1):
        for d in (0, 1):
            s = a[dec, d]
            base = int(s[:, 0, 0].min())
            print("%s dir %d, workgroup 0, last steady iteration (cycles of s_memtime; region order r0 r1 r2 r3):" % ("decoder" if dec else "encoder", d))
            print("  wave | per region: start (after the barrier, relative)  %s  LDS drain  wait at the barrier"
                  % ("gi wait  slot stream + tail" if precision == "fp32x3" else "slot stream  h stores + outbound + VMEM wait"))
            for v in range(8):
                parts = []
                for r in range(4):
                    t = [int(z) for z in s[v, r]]
                    nxt = int(s[v, r + 1, 0]) if r < 3 else None
                    parts.append("%6d %5d %5d %4d %5s" % (t[0] - base, t[1] - t[0], t[2] - t[1], t[3] - t[2], "-" if nxt is None else str(nxt - t[3])))
                print("  %d    | %s" % (v, " | ".join(parts)))
    if precision == "fp32x3" and hasattr(lib, "helen_debug_x3f"):
        # region r0 of the same iteration: a stamp in front of every fifth slot-step of the stream (MFMA i + gate slot i)
        lib.helen_debug_x3f(buf)
        f = torch.tensor(list(buf), dtype=torch.int64).reshape(2, 2, 8, 16)
        for dec in (0, 1):
            print("%s dir 0: cycles between every fifth slot-step of region r0 (steps 0-5, 5-10, ... 70-75)" % ("decoder" if dec else "encoder"))
            for v in range(8):
                t = [int(z) for z in f[dec, 0, v]]
                print("  %d    | start %5d | %s" % (v, t[0] - int(a[dec, 0, v, 0, 0]), " ".join("%4d" % (t[k + 1] - t[k]) for k in range(14))))
    if precision == "bf16" and hasattr(lib, "helen_debug_bf16f"):
        # region r0: a stamp in front of every fourth slot-step (0, 4, ... 44), then after the loop's last slot (index 12),
        # after the new h is in LDS and the outbound stores are issued (13); stamp 2 of the region = after the VMEM wait
        lib.helen_debug_bf16f(buf)
        f = torch.tensor(list(buf), dtype=torch.int64).reshape(2, 2, 8, 16)
        for dec in (0, 1):
            print("%s dir 0, region r0: start -> slot 0 | cycles per four slot-steps | loop end -> h stored | -> stores issued | -> VMEM wait done"
                  % ("decoder" if dec else "encoder"))
            for v in range(8):
                t = [int(z) for z in f[dec, 0, v]]
                last = max(k for k in range(12) if t[k])
                print("  %d    | %5d | %s | %4d %4d %4d" % (v, t[0] - int(a[dec, 0, v, 0, 0]), " ".join("%4d" % (t[k + 1] - t[k]) for k in range(last)),
                                                         t[12] - t[last], t[13] - t[12], int(a[dec, 0, v, 0, 2]) - t[13]))


if __name__ == "__main__":
    main()
