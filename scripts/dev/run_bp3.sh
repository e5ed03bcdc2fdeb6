#!/bin/bash
# Developer probe (run ON the GPU box): timing only, for probe builds whose results are garbage
for n in "$@"; do
  echo "== $n"
  HELEN_BF16_PAIR=1 HELEN_HIP_LIB=$PWD/build/lib_bp_$n.so timeout 300 python scripts/quick_bench.py --windows 4096 --iters 5 --precision bf16 2>&1 | grep -E "gru"
done
