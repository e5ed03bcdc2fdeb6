#!/bin/bash
# Developer probe (run ON the GPU box): bf16 mode at batch 512 with the decoder on the pair kernel (HELEN_BF16_IL=10, the
# default) and on the interleaved kernel (11), for the default library and every build/lib_<name>.so given.
#   scripts/dev/ab_bf16_il.sh ad5 ad6
mkdir -p gpurun_out/ab_il
for lib in default "$@"; do
  for il in 10 11; do
    if [ $lib = default ]; then unset HELEN_HIP_LIB; else export HELEN_HIP_LIB=$PWD/build/lib_$lib.so; fi
    HELEN_BF16_IL=$il python bench.py --precision bf16 --batch 512 --no-cpu-baseline --no-host-path --no-margins --e2e 0 \
        > gpurun_out/ab_il/${lib}_$il.json 2> gpurun_out/ab_il/${lib}_$il.err
    python - <<EOF
import json
d=json.loads(open("gpurun_out/ab_il/${lib}_$il.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("%-8s IL=%s  %.0f windows/s  enc %.4f ms  dec %.4f ms  label_identity %s" % ("$lib", "$il", d["value"], r["avg_launch_ms_encoder"], r["avg_launch_ms_decoder"], d["precision_check"]["label_identity"]))
EOF
  done
done
