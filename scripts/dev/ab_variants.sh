#!/bin/bash
# Developer probe (run ON the GPU box): per-kernel times of helen_polish_batch (4096 windows) for every
# library variant build/lib_*.so given on the command line (built side by side with different -D flags).
#   scripts/dev/ab_variants.sh base noslp ...   -> gpurun_out/ab_<name>.log
mkdir -p gpurun_out
for v in "$@"; do
    HELEN_HIP_LIB=$PWD/build/lib_$v.so python scripts/quick_bench.py --windows 4096 --iters 5 > gpurun_out/ab_$v.log 2>&1
    echo "== $v"; grep -E "n=|gru_|gemm_|heads|pack" gpurun_out/ab_$v.log
done
