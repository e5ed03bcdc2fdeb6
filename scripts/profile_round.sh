#!/bin/bash
# Round profile of the default bench (run ON the GPU box, from the repo root):
#   1. the bench line itself                       -> gpurun_out/bench.json
#   2. rocprofv3 --kernel-trace --stats            -> gpurun_out/prof/  (per-kernel durations)
#   3. three PMC passes, each with --kernel-trace only (FETCH_SIZE, WRITE_SIZE, SQ counters)
#                                                  -> gpurun_out/pmc_*/
# Then, back in the build container:  PMC_WINDOWS=8192 python scripts/pmc_summary.py rNN ; copy the stats csv to profiles/.
set -u
R=$(pwd)
mkdir -p $R/gpurun_out
python bench.py > $R/gpurun_out/bench.json 2> $R/gpurun_out/bench.err
# (the opt-in arithmetic modes are in the default line's `modes` object; scripts/profile_mode.sh profiles one of them)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o p -- \
    python $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-host-path --no-margins --no-modes --no-traffic --e2e 0 > $R/gpurun_out/prof.log 2>&1
# counter passes: TWO device calls of 4,096 windows and nothing else on the device (scripts/pmc_one_call.py), one counter set
# per pass, --kernel-trace only (never with --sys-trace / hip / hsa domains: see the harness notes)
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o p -- \
        python $R/scripts/pmc_one_call.py fp32 4096 2 > $R/gpurun_out/pmc_$c.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \
    SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_sq -o p -- \
    python $R/scripts/pmc_one_call.py fp32 4096 2 > $R/gpurun_out/pmc_sq.log 2>&1
cd $R
cat gpurun_out/bench.json
find gpurun_out -name "*.csv" | head -20
