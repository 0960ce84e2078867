#!/usr/bin/env python3
"""Per-launch durations of the kd-tree build kernels from a rocprofv3 results database.
    python tests/prof_kd_levels.py gpurun_out/kdprof/kd_results.db"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, grid_x, workgroup_x from kernels where name like '%k_kd%' order by start").fetchall()
total = 0.0
for name, start, end, grid, wg in rows:
    ms = (end - start) / 1e6
    total += ms
    print(f"{name.split('(')[0][:48]:48s} {ms:9.3f} ms  {grid // wg:8d} x {wg}")
print(f"total {total:.3f} ms over {(rows[-1][2] - rows[0][1]) / 1e6:.3f} ms")
