#!/bin/bash
# The N = 4 and N = 8 bites of the SCALE rehearsal, the second only if the first stayed under 70 % of the memory cgroup.
# Writes gpurun_out/scale_sd/*; see scripts/dev/scale_bite.sh.
set -u
LIM=$(cat /sys/fs/cgroup/memory.max 2>/dev/null || echo max)
[ "$LIM" = max ] && LIM=$(awk '/MemTotal/{print $2*1024}' /proc/meminfo)
for N in "$@"; do
    scripts/dev/scale_bite.sh $N
    PEAK=$(awk '!/^#/{if($2>m)m=$2} END{print m*1048576}' gpurun_out/scale_sd/n$N.watch)
    echo "N=$N peak $PEAK of limit $LIM"
    if [ $(python3 -c "print(int($PEAK > 0.7 * $LIM))") = 1 ]; then echo "peak above 70 % of the limit: stopping before the next N"; break; fi
done
