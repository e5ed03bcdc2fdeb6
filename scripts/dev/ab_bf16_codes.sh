#!/bin/bash
# Developer probe (run ON the GPU box): bf16 mode at batch 512 for a list of HELEN_BF16_IL codes (encoder digit, decoder
# digit: 0 pair, 1 interleaved (the default for both)), optionally with another library:  LIB=name scripts/dev/ab_bf16_codes.sh 11 00 10
mkdir -p gpurun_out/ab_il
if [ -n "$LIB" ]; then export HELEN_HIP_LIB=$PWD/build/lib_$LIB.so; fi
for il in "$@"; do
    HELEN_BF16_IL=$il python bench.py --precision bf16 --batch 512 --no-cpu-baseline --no-host-path --no-margins --e2e 0 \
        > gpurun_out/ab_il/${LIB:-default}_$il.json 2> gpurun_out/ab_il/${LIB:-default}_$il.err
    python - <<EOF
import json
try:
    d=json.loads(open("gpurun_out/ab_il/${LIB:-default}_$il.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("%-8s IL=%s  %.0f windows/s  enc %.4f ms  dec %.4f ms  label_identity %s" % ("${LIB:-default}", "$il", d["value"], r["avg_launch_ms_encoder"], r["avg_launch_ms_decoder"], d["precision_check"]["label_identity"]))
except Exception as e:
    print("${LIB:-default} IL=$il failed:", e); print(open("gpurun_out/ab_il/${LIB:-default}_$il.err").read()[-1500:])
EOF
done
