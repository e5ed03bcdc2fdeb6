#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (gpurun_out/pmc_*/p_counter_collection.csv) into
profiles/<tag>_pmc_summary.json: per-kernel HBM bytes per launch and MFMA utilisation.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are
in KiB and collected in separate passes; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide
(16 B/lane) coalesced reads, so it is doubled.  Both are calibrated here on kernels whose byte counts
are known exactly (pack_images writes 1,572,864,000 B per 4096 windows; the recurrence reads gi).
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs).
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
# optional: the directory holding pmc_FETCH_SIZE / pmc_WRITE_SIZE / pmc_sq (default gpurun_out) and a note on the command
src = os.path.join(ROOT, sys.argv[2]) if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out")
extra = sys.argv[3] if len(sys.argv) > 3 else ""


_demangled = {}


def demangle(name):
    """rocprofv3 leaves some template instantiations mangled (_ZN5helen26gru_fused_bf16_pair_kernelILi3ELb0EE...: the
    bf16 vector type in the signature is beyond its demangler).  Enough of the Itanium scheme for this library's
    kernels: namespace helen, the name, integer / bool template arguments."""
    import re
    m = re.match(r"_ZN5helen(\d+)", name)
    if not m:
        return name
    if name not in _demangled:
        n = int(m.group(1))
        base = name[m.end():m.end() + n]
        rest = name[m.end() + n:]
        args = []
        if rest.startswith("I"):
            for kind, val in re.findall(r"L([ib])(n?\d+)E", rest[1:rest.index("EE") + 1] if "EE" in rest else ""):
                args.append(("true" if val != "0" else "false") if kind == "b" else val.replace("n", "-"))
        _demangled[name] = "helen::" + base + ("<" + ", ".join(args) + ">" if args else "")
    return _demangled[name]


def agg(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = demangle(r["Kernel_Name"]).split("(")[0].replace("void ", "").strip()
        d[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r.get("End_Timestamp"):
            d[name]["_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return d


fetch = agg(os.path.join(src, "pmc_FETCH_SIZE", "p_counter_collection.csv"))
write = agg(os.path.join(src, "pmc_WRITE_SIZE", "p_counter_collection.csv"))
sq = agg(os.path.join(src, "pmc_sq", "p_counter_collection.csv"))
windows = int(os.environ.get("PMC_WINDOWS", "0"))      # windows the profiled command processed in all (pmc_one_call.py: calls x n)
out = {"command": ("rocprofv3 --kernel-trace --pmc <counters> -- python scripts/pmc_one_call.py " + (extra or "fp32 4096 2")) if windows else
                  ("rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --steps 1 --warmup 0 "
                   "--no-cpu-baseline --no-host-path --e2e 0" + ((" " + extra) if extra else "")),
       "notes": "FETCH_SIZE doubled (gfx950 wide-read correction); sizes in bytes per launch",
       "kernels": {}}
for k in sorted(sq):
    if not k.startswith("helen::"):
        continue
    mean = lambda d, c: sum(d[k][c]) / len(d[k][c]) if d.get(k, {}).get(c) else None  # noqa: E731
    f, w = mean(fetch, "FETCH_SIZE"), mean(write, "WRITE_SIZE")
    gui = mean(sq, "GRBM_GUI_ACTIVE")
    mf = mean(sq, "SQ_VALU_MFMA_BUSY_CYCLES")
    wave = mean(sq, "SQ_WAVE_CYCLES")
    ns = mean(sq, "_ns")
    out["kernels"][k] = {
        # the chip clocks to its power budget: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / kernel duration
        "effective_clock_GHz": round(gui / 8 / ns, 3) if gui and ns else None,
        "duration_ms_in_this_pass": round(ns / 1e6, 4) if ns else None,
        "launches_profiled": len(sq[k]["GRBM_GUI_ACTIVE"]),
        "hbm_read_bytes_per_launch": None if f is None else int(2 * f * 1024),
        "hbm_write_bytes_per_launch": None if w is None else int(w * 1024),
        "hbm_bytes_per_launch": None if f is None or w is None else int((2 * f + w) * 1024),
        "mfma_util": round(mf / (gui / 8 * 1024), 4) if gui else None,
        "wait_inst_any_frac": round(mean(sq, "SQ_WAIT_INST_ANY") / wave, 4),
        "wait_any_frac": round(mean(sq, "SQ_WAIT_ANY") / wave, 4),
        "active_inst_frac": round(mean(sq, "SQ_ACTIVE_INST_ANY") / wave, 4),
        "lds_bank_conflict_cycles": mean(sq, "SQ_LDS_BANK_CONFLICT"),
    }
if windows:
    # every launch of every kernel of the library in the pass, read + write, over the windows the command processed
    total = 0.0
    for k in out["kernels"]:
        if fetch.get(k, {}).get("FETCH_SIZE"):
            total += 2 * 1024 * sum(fetch[k]["FETCH_SIZE"])
        if write.get(k, {}).get("WRITE_SIZE"):
            total += 1024 * sum(write[k]["WRITE_SIZE"])
    out["windows_in_the_pass"] = windows
    out["call_bytes_per_window"] = int(total / windows)
    out["call_bytes_over_algorithmic"] = round(total / windows / 92000.0, 1)
path = os.path.join(ROOT, "profiles", tag + "_pmc_summary.json")
json.dump(out, open(path, "w"), indent=1)
print(open(path).read())
