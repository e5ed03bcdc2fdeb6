#!/bin/bash
# Developer probe (run ON the GPU box): gru_fused_bf16_pair_kernel against gru_fused_bf16_kernel -- results bit for
# bit, per-kernel times, and (build/lib_bpt.so = -DHELEN_BP_TIMING) where a wave's cycles go.
export HELEN_HIP_LIB=$PWD/build/lib_bp.so HELEN_AB_PRECISION=bf16
HELEN_BF16_PAIR=0 python scripts/dev/ab_equal.py save /tmp/a.pt 4096 || exit 1
HELEN_BF16_PAIR=1 timeout 300 python scripts/dev/ab_equal.py cmp /tmp/a.pt 4096
HELEN_BF16_PAIR=0 python scripts/dev/ab_equal.py save /tmp/b.pt 4080 4080 && HELEN_BF16_PAIR=1 timeout 300 python scripts/dev/ab_equal.py cmp /tmp/b.pt 4080 4080
for p in 0 1 1; do echo "== pair $p"; HELEN_BF16_PAIR=$p timeout 300 python scripts/quick_bench.py --windows 4096 8192 --iters 5 --precision bf16 2>&1 | grep -E "n=|gru|heads|pack"; done
if [ -f build/lib_bpt.so ]; then
  HELEN_HIP_LIB=$PWD/build/lib_bpt.so HELEN_BF16_PAIR=1 timeout 300 python scripts/quick_bench.py --windows 4096 --iters 1 --precision bf16 2>&1 | grep "bf16 pair" | tail -32 | sort
fi
