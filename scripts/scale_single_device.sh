#!/bin/bash
# First-contact insurance for the driver's 8-GPU SCALE run (run ON the one-GPU box, from the repo root): the EXACT command
# shapes the driver uses -- `bench.py --gpus N --steps 20 --warmup 5` for N = 1 plainly and N = 2, 4, 8 under
# `python -m torch.distributed.run` -- with every default leg on, all ranks sharing cuda:0 (--single-device).
# Writes gpurun_out/scale_single_device.json: per N the bench line, its wall-clock seconds and stderr's shrink messages.
set -u
R=$(pwd)
O=$R/gpurun_out/scale_sd
mkdir -p $O
for N in 1 2 4 8; do
    t0=$(date +%s.%N)
    if [ $N = 1 ]; then
        timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/n$N.json 2> $O/n$N.err
    else
        timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
            bench.py --gpus $N --steps 20 --warmup 5 --single-device > $O/n$N.json 2> $O/n$N.err
    fi
    rc=$?
    t1=$(date +%s.%N)
    echo "$N $rc $(echo "$t1 - $t0" | bc)" >> $O/walls.txt
    echo "N=$N rc=$rc wall=$(echo "$t1 - $t0" | bc) s"
done
python - <<PY
import json, os
O = "$O"
out = {"what": "bench.py --gpus N --steps 20 --warmup 5 (all default legs) in the driver's command shapes, ranks sharing ONE MI355X "
               "(--single-device); N = 1 is the plain command", "runs": []}
for ln in open(os.path.join(O, "walls.txt")):
    n, rc, wall = ln.split()
    line = None
    for l in open(os.path.join(O, "n%s.json" % n)):
        l = l.strip()
        if l.startswith("{"):
            line = json.loads(l)
    err = open(os.path.join(O, "n%s.err" % n)).read()
    notes = [l for l in err.splitlines() if "shrink" in l.lower() or "/dev/shm" in l]
    out["runs"].append({"n": int(n), "rc": int(rc), "wall_seconds": round(float(wall), 1), "stderr_notes": notes[:12], "line": line})
json.dump(out, open(os.path.join("$R", "gpurun_out", "scale_single_device.json"), "w"), indent=1)
for r in out["runs"]:
    l = r["line"] or {}
    e = l.get("end_to_end") or {}
    print(r["n"], r["rc"], r["wall_seconds"], l.get("value"), e.get("value"), e.get("windows"), r["stderr_notes"][:2])
PY
