#!/usr/bin/env python3
"""Kernel timeline of the last frame in a rocprofv3 --kernel-trace output directory: tests/prof_trace_timeline.py DIR [N]"""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
stats = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(stats)))[:8]:
    print(f"{r['Name'][:44]:44s} calls {int(r['Calls']):5d} total {float(r['TotalDurationNs'])/1e6:9.3f} ms avg {float(r['AverageNs'])/1e6:8.3f} ms {r['Percentage']:>6s} %")
trace = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
ks = sorted(((r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp']), r['VGPR_Count'], r['Scratch_Size'], r['Grid_Size_X']) for r in csv.DictReader(open(trace))), key=lambda x: x[1])
gi = [i for i, k in enumerate(ks) if 'generate' in k[0] or 'k_paths' in k[0]]
if gi:
    t0 = ks[gi[-1]][1]
    for k in ks[gi[-1]:gi[-1] + n]:
        if 'rocclr' in k[0]:
            continue
        print(f"{k[0][:34]:34s} start {(k[1]-t0)/1e6:8.3f} dur {(k[2]-k[1])/1e6:8.3f} vgpr {k[3]} scratch {k[4]} grid {k[5]}")
    last = [k for k in ks[gi[-1]:] ]
    print("frame span ms", (last[-1][2] - t0) / 1e6)
