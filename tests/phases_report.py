#!/usr/bin/env python3
"""Phase shares of the ray-list kernel from a bench.py line of an RT_PROFILE_PHASES build: tests/phases_report.py FILE"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ph = d["phase_cycles"]
print("ms/frame", round(d["ms_per_step"], 2))
tot = sum(ph[0:4]) + ph[12]
for i, nm in enumerate(("refill", "scene", "walk", "leaf")):
    print(f"{nm:7s} {100*ph[i]/tot:5.1f} % of wave time, {ph[4+i]:12d} executions, {ph[8+i]/max(1,ph[4+i]):5.1f} lanes, {ph[i]/max(1,ph[4+i])*10:7.0f} ns each")
print(f"other   {100*ph[12]/tot:5.1f} %")
