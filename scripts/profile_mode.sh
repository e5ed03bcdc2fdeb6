#!/bin/bash
# Profile ONE arithmetic mode of the bench (run ON the GPU box, from the repo root):
#   scripts/profile_mode.sh fp32x3|bf16 [outdir under gpurun_out]
# kernel trace + stats, then the three PMC passes (each with --kernel-trace only), like profile_round.sh.
# Back in the build container:  python scripts/pmc_summary.py rNN_<mode> gpurun_out/<outdir> "--precision <mode>"
set -u
R=$(pwd)
P=${1:-fp32x3}
O=$R/gpurun_out/${2:-prof_$P}
B=$( [ $P = bf16 ] && echo "--batch 512" )
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--precision $P $B --no-cpu-baseline --no-host-path --no-margins --no-modes --no-traffic --e2e 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- \
    python $R/bench.py --steps 16 --warmup 2 $COMMON > $O/prof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- \
        python $R/bench.py --steps 1 --warmup 0 $COMMON > $O/pmc_$c.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \
    SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq -o p -- \
    python $R/bench.py --steps 1 --warmup 0 $COMMON > $O/pmc_sq.log 2>&1
cd $R
find $O -name "*stats*.csv" | head; tail -2 $O/prof.log
