#!/bin/bash
# Samples what can take a GPU box down while a command runs (host RAM against the memory cgroup, tmpfs, page-locked
# memory, VRAM of every card, processes), once a second, into the file given as $1; a header block says what the box
# grants.  Usage: scripts/dev/watch_box.sh gpurun_out/watch.log & W=$!; <command>; kill $W
out=${1:-gpurun_out/watch.log}
mkdir -p "$(dirname "$out")"
cg=/sys/fs/cgroup
{
  echo "# $(date -u +%FT%TZ) host $(hostname) nproc $(nproc) kernel $(uname -r)"
  echo "# memory.max $(cat $cg/memory.max 2>/dev/null) memory.high $(cat $cg/memory.high 2>/dev/null) memory.swap.max $(cat $cg/memory.swap.max 2>/dev/null)"
  echo "# v1 limit $(cat $cg/memory/memory.limit_in_bytes 2>/dev/null)"
  echo "# cpu.max $(cat $cg/cpu.max 2>/dev/null) cpuset $(cat $cg/cpuset.cpus.effective 2>/dev/null) pids.max $(cat $cg/pids.max 2>/dev/null)"
  echo "# ulimit -l $(ulimit -l) -n $(ulimit -n) -u $(ulimit -u)"
  grep -E 'MemTotal|MemAvailable|Unevictable|Mlocked|Shmem:|SwapTotal' /proc/meminfo | sed 's/^/# /'
  df -h /dev/shm /tmp . 2>/dev/null | sed 's/^/# /'
  echo "# memory.events $(tr '\n' ' ' < $cg/memory.events 2>/dev/null)"
  echo "# columns: t cg_current_MB mem_avail_MB shmem_MB unevictable_MB shm_used_MB vram_used_MB python_rss_MB nprocs oom_kill load1"
} > "$out"
t0=$(date +%s)
while true; do
  now=$(( $(date +%s) - t0 ))
  cur=$(( $(cat $cg/memory.current 2>/dev/null || echo 0) / 1048576 ))
  avail=$(awk '/MemAvailable/{print int($2/1024)}' /proc/meminfo)
  shmem=$(awk '/^Shmem:/{print int($2/1024)}' /proc/meminfo)
  unev=$(awk '/^Unevictable:/{print int($2/1024)}' /proc/meminfo)
  shmu=$(df -m /dev/shm 2>/dev/null | awk 'NR==2{print $3}')
  vram=0
  for f in /sys/class/drm/card*/device/mem_info_vram_used; do [ -r "$f" ] && vram=$(( vram + $(cat $f) / 1048576 )); done
  rss=$(ps -eo rss,comm | awk '/python/{s+=$1} END{print int(s/1024)}')
  np=$(ps -e --no-headers | wc -l)
  oom=$(awk '/^oom_kill /{print $2}' $cg/memory.events 2>/dev/null)
  l1=$(cut -d' ' -f1 /proc/loadavg)
  echo "$now $cur $avail $shmem $unev $shmu $vram $rss $np ${oom:-?} $l1" >> "$out"
  sleep 1
done
