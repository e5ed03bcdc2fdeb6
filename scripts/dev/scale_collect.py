#!/usr/bin/env python3
"""Collect the bites of the SCALE rehearsal (scripts/dev/scale_bite.sh, one value of N per run, all ranks on the one
MI355X) into one JSON document: per N the bench line, the wall clock of the whole command, and what
scripts/dev/watch_box.sh saw of the box meanwhile (peak memory of the cgroup, of /dev/shm, oom kills).
Usage: scale_collect.py gpurun_out/scale_sd > profiles/rNN_scale_single_device.json"""
import json
import os
import sys

d = sys.argv[1]
walls = {}
for ln in open(os.path.join(d, "walls.txt")):
    tag, rc, wall = ln.split()
    walls[tag] = (int(rc), float(wall) if wall else None)
out = {"what": "bench.py --gpus N --steps 20 --warmup 5 with every default leg, in the driver's command shapes (N = 1 plainly, N > 1 "
               "under python -m torch.distributed.run), all ranks on ONE MI355X (--single-device); one gpurun bite per N "
               "(scripts/dev/scale_bite.sh), the box sampled once a second beside it (scripts/dev/watch_box.sh)", "runs": []}
for tag in sorted(walls, key=lambda t: (int(t.split("_")[0]), t)):
    line = None
    p = os.path.join(d, "n%s.json" % tag)
    if os.path.exists(p):
        for l in open(p):
            if l.startswith("{"):
                line = json.loads(l)
    header, rows = [], []
    for l in open(os.path.join(d, "n%s.watch" % tag)):
        (header if l.startswith("#") else rows).append(l.split())
    rows = [[float(x) if x not in ("?",) else 0.0 for x in r] for r in rows if len(r) >= 10]
    limit = next((int(h[2]) for h in header if len(h) > 2 and h[1] == "memory.max" and h[2].isdigit()), None)
    watch = {"samples": len(rows), "memory_cgroup_limit_MB": None if limit is None else limit >> 20,
             "peak_cgroup_MB": max(r[1] for r in rows), "peak_dev_shm_MB": max(r[5] for r in rows),
             "peak_cgroup_minus_shm_MB": max(r[1] - r[5] for r in rows), "peak_python_rss_MB": max(r[7] for r in rows),
             "peak_processes": max(r[8] for r in rows), "oom_kill": rows[-1][9]}
    e = (line or {}).get("end_to_end") or {}
    out["runs"].append({"n": int(tag.split("_")[0]), "tag": tag, "rc": walls[tag][0], "wall_seconds": walls[tag][1], "box": watch,
                        "value": (line or {}).get("value"), "per_rank": (line or {}).get("per_rank_windows_per_s"),
                        "host_path": ((line or {}).get("host_path") or {}).get("value"),
                        "end_to_end": {k: e.get(k) for k in ("value", "windows", "seconds", "polish_seconds", "n_ranks", "skipped", "error")},
                        "line": line})
json.dump(out, sys.stdout, indent=1)
for r in out["runs"]:
    sys.stderr.write("N=%d%s rc %d wall %s s value %s e2e %s peak cgroup %.0f MB shm %.0f MB oom %s\n"
                     % (r["n"], r["tag"][len(str(r["n"])):], r["rc"], r["wall_seconds"], r["value"], r["end_to_end"].get("value"),
                        r["box"]["peak_cgroup_MB"], r["box"]["peak_dev_shm_MB"], r["box"]["oom_kill"]))
