#!/bin/bash
# bf16 mode (BASELINE.json configs[3]) profile, run ON the GPU box from the repo root:
#   kernel stats (rocprofv3 --kernel-trace --stats) and three PMC passes of
#   `bench.py --precision bf16 --batch 512` -> gpurun_out/bf16/
# Back in the build container:  python scripts/pmc_summary.py r03_bf16 gpurun_out/bf16 "--precision bf16 --batch 512"
set -u
R=$(pwd)
O=$R/gpurun_out/bf16
mkdir -p $O
ARGS="--precision bf16 --batch 512 --no-cpu-baseline --no-host-path --no-margins --no-traffic --e2e 0"
python bench.py $ARGS > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- \
    python $R/bench.py $ARGS --steps 16 --warmup 2 > $O/prof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- \
        python $R/bench.py $ARGS --steps 1 --warmup 0 > $O/pmc_$c.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \
    SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $O/pmc_sq -o p -- \
    python $R/bench.py $ARGS --steps 1 --warmup 0 > $O/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_MFMA \
    SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_sq2 -o p -- \
    python $R/bench.py $ARGS --steps 1 --warmup 0 > $O/pmc_sq2.log 2>&1
cd $R
cat $O/bench.json
find $O -name "*.csv" | head -20
