#!/bin/bash
# One bite of the driver's SCALE rehearsal on the one-GPU box: bench.py in the driver's command shape for ONE value of N,
# all ranks on cuda:0 (--single-device), with scripts/dev/watch_box.sh sampling the box beside it.
# Usage: scripts/dev/scale_bite.sh N [extra bench.py flags]   -> gpurun_out/scale_sd/n<N><tag>.{json,err,watch}
set -u
N=$1; shift
TAG=${TAG:-}
O=gpurun_out/scale_sd
mkdir -p $O
scripts/dev/watch_box.sh $O/n$N$TAG.watch & W=$!
t0=$(date +%s.%N)
if [ $N = 1 ]; then
    timeout ${LIMIT:-1200} python bench.py --gpus 1 --steps 20 --warmup 5 "$@" > $O/n$N$TAG.json 2> $O/n$N$TAG.err
else
    timeout ${LIMIT:-1200} python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) \
        bench.py --gpus $N --steps 20 --warmup 5 --single-device "$@" > $O/n$N$TAG.json 2> $O/n$N$TAG.err
fi
rc=$?
t1=$(date +%s.%N)
kill $W 2>/dev/null
echo "$N$TAG $rc $(python3 -c "print(round($t1 - $t0, 1))")" >> $O/walls.txt
echo "N=$N$TAG rc=$rc wall=$(python3 -c "print(round($t1 - $t0, 1))") s; peak cgroup MB $(awk '!/^#/{if($2>m)m=$2} END{print m}' $O/n$N$TAG.watch), peak shm MB $(awk '!/^#/{if($6>m)m=$6} END{print m}' $O/n$N$TAG.watch), peak vram MB $(awk '!/^#/{if($7>m)m=$7} END{print m}' $O/n$N$TAG.watch), oom_kill $(awk '!/^#/{o=$10} END{print o}' $O/n$N$TAG.watch)"
dmesg 2>/dev/null | tail -30 > $O/n$N$TAG.dmesg
head -c 600 $O/n$N$TAG.json; echo
tail -5 $O/n$N$TAG.err
