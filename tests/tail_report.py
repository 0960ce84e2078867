#!/usr/bin/env python3
"""tests/tail_report.py [bench.py arguments]: how long the waves of the path kernel's launch sit idle at its end.
Needs a library built with -DRT_PROFILE_TAIL (raytracer_amd/csrc/build_variant.sh tail -DRT_PROFILE_TAIL; RT_LIB=...)."""
import json, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"] + sys.argv[1:],
                     capture_output=True, text=True).stdout
d = json.loads(out.strip().splitlines()[-1])
ph = d["phase_cycles"]
t0, t1, total, waves = ph[12], ph[13], ph[14], ph[15]
span = (t1 - t0) / 100e3  # 100 MHz clock -> ms
mean_busy = total / waves / 100e3
print(f"{' '.join(sys.argv[1:]) or 'whole frame':24s} launch {span:7.2f} ms, a wave is busy {mean_busy:7.2f} ms on average: "
      f"{span - mean_busy:5.2f} ms ({100 * (1 - mean_busy / span):4.1f} %) of the launch it idles; kernel by events {d['roofline']['avg_launch_ms']:.2f} ms")
