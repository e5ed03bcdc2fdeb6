#!/bin/bash
# Developer probe (run ON the GPU box): variants of gru_fused_bf16_pair_kernel built side by side as build/lib_bp_<name>.so
export HELEN_AB_PRECISION=bf16 HELEN_BF16_PAIR=1
HELEN_HIP_LIB=$PWD/build/lib_bp.so python scripts/dev/ab_equal.py save /tmp/a.pt 4096 || exit 1
for n in "$@"; do
  echo "== $n"
  HELEN_HIP_LIB=$PWD/build/lib_bp_$n.so timeout 300 python scripts/dev/ab_equal.py cmp /tmp/a.pt 4096 | grep -c EQUAL
  HELEN_HIP_LIB=$PWD/build/lib_bp_$n.so timeout 300 python scripts/quick_bench.py --windows 4096 --iters 5 --precision bf16 2>&1 | grep -E "n=|gru"
  if [ -f build/lib_bp_${n}t.so ]; then
    HELEN_HIP_LIB=$PWD/build/lib_bp_${n}t.so timeout 300 python scripts/quick_bench.py --windows 4096 --iters 1 --precision bf16 2>&1 | grep "bf16 pair" | tail -32 | sort | grep "dir 0"
  fi
done
