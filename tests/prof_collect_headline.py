#!/usr/bin/env python3
"""tests/prof_collect_headline.py TAG SQTAG NAME [KERNEL]: condense gpurun_out/TAG (tests/prof_round.sh) and
gpurun_out/SQTAG (tests/prof_sq.sh) of the default bench command into profiles/NAME_kernel_stats.csv,
NAME_bench_under_rocprof.json and NAME_pmc_KERNEL.json (what bench.py's roofline.traffic reads).  KERNEL is the
dominant kernel of the frame: k_rays (wavefront pipeline, the default for scenes with meshes) or k_paths."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, sqtag, name = sys.argv[1:4]
KERNEL = sys.argv[4] if len(sys.argv) > 4 else "k_rays"
src = os.path.join(ROOT, "gpurun_out", tag)
prof = os.path.join(ROOT, "profiles")
shutil.copy(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)[0], os.path.join(prof, f"{name}_kernel_stats.csv"))
line = [l for l in open(os.path.join(src, "bench_under_trace.json")).read().splitlines() if l.startswith("{")][-1]
open(os.path.join(prof, f"{name}_bench_under_rocprof.json"), "w").write(line + "\n")


def sums(directory, want_timed=True):
    """counter sums over the dispatches of the timed frame's path kernel (the counting variant is k_paths<true, ...>)"""
    out, kernel, n = {}, None, set()
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if KERNEL + "<false" not in k:
                continue
            kernel = k.split("(")[0].replace("void ", "")
            out[r["Counter_Name"]] = out.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            n.add((f, r["Dispatch_Id"]))
    return out, kernel, n


mem = {}
launches = 1
for sub in ("fetch", "write", "cache", "rdreq", "wrreq"):
    d, kernel, disp = sums(os.path.join(src, sub))
    mem.update(d)
    launches = max(launches, len(disp))
sq, _, _ = sums(os.path.join(ROOT, "gpurun_out", sqtag))
out = {
    "kernel": kernel,
    "workload": json.loads(line)["config"]["workload"],
    "launches_profiled": launches,
    "command": "rocprofv3 --pmc <one block per pass> -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline (tests/prof_round.sh, tests/prof_sq.sh)",
    "memory_side": {
        "read_bytes": 2.0 * mem["FETCH_SIZE"] * 1024,
        "write_bytes": mem["WRITE_SIZE"] * 1024,
        "FETCH_SIZE_KB_raw": mem["FETCH_SIZE"], "WRITE_SIZE_KB_raw": mem["WRITE_SIZE"],
        "note": "reads: every request is 128 B (TCC_EA0_RDREQ_128B), so bytes = 2 x FETCH_SIZE, the gfx950 correction; writes: 64-B and 32-B requests",
    },
    "tcc": {k: v for k, v in mem.items() if k.startswith("TC")},
    "sq": sq,
}
json.dump(out, open(os.path.join(prof, f"{name}_pmc_{KERNEL}.json"), "w"), indent=1)
print(json.dumps(out["memory_side"], indent=1))
print("launches in one frame", launches)
print("lanes per VALU instruction", sq["SQ_THREAD_CYCLES_VALU"] / sq["SQ_ACTIVE_INST_VALU"])
