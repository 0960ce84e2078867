#!/usr/bin/env python3
"""tests/prof_collect.py TAG NAME: condense gpurun_out/TAG (written by tests/prof_workload.sh) into
profiles/NAME_kernel_stats.csv (rocprofv3 --kernel-trace --stats) and profiles/NAME_pmc.json (counter sums per kernel)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", tag)
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{name}_kernel_stats.csv"))
bench = os.path.join(src, "bench_under_trace.json")
if os.path.exists(bench):
    lines = [l for l in open(bench).read().splitlines() if l.startswith("{")]
    if lines:
        open(os.path.join(ROOT, "profiles", f"{name}_bench_under_rocprof.json"), "w").write(lines[-1] + "\n")
sums = {}
for f in glob.glob(os.path.join(src, "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith("rtd::"):
            continue
        d = sums.setdefault(k, {})
        d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        d.setdefault("_dispatches", set()).add((f, row["Dispatch_Id"]))
out = {}
for k, d in sums.items():
    disp = d.pop("_dispatches")
    passes = len({f for f, _ in disp})
    o = dict(d)
    o["dispatches_per_pass"] = len(disp) // max(1, passes)
    if "FETCH_SIZE" in o:  # KB; on gfx950 every memory-side read request is 128 B and the counter assumes 64: double it
        o["read_bytes"] = 2.0 * o["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in o:
        o["write_bytes"] = o["WRITE_SIZE"] * 1024
    out[k] = o
json.dump({"source": f"gpurun_out/{tag} (tests/prof_workload.sh: one rocprofv3 --pmc pass per counter block, bench.py --steps 1 --warmup 0)",
           "note": "sums over all dispatches of the kernel in one pass: the counting frame plus one timed frame",
           "kernels": out}, open(os.path.join(ROOT, "profiles", f"{name}_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
