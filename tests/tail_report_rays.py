#!/usr/bin/env python3
"""tests/tail_report_rays.py [bench.py arguments]: how the waves of ONE ray launch of the wavefront pipeline spend the
launch - when they start, when they find the ray list empty, when they end.  Needs a library built with
-DRT_PROFILE_TAIL=<bounce> (raytracer_amd/csrc/build_variant.sh tail0 -DRT_PROFILE_TAIL=0; RT_LIB=...)."""
import json, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline"] + sys.argv[1:],
                     capture_output=True, text=True).stdout
d = json.loads(out.strip().splitlines()[-1])
ph = d["phase_cycles"]
t0, t1, s_start, s_exh, s_end, waves, max_exh, min_exh = ph[:8]
ms = lambda c: c / 100e3
print(f"{' '.join(sys.argv[1:]) or 'whole frame':28s} waves {waves}: launch {ms(t1 - t0):6.3f} ms | mean start +{ms(s_start / waves - t0):6.3f} | list empty: first +{ms(min_exh - t0):6.3f} "
      f"mean +{ms(s_exh / waves - t0):6.3f} last +{ms(max_exh - t0):6.3f} | end: mean +{ms(s_end / waves - t0):6.3f} last +{ms(t1 - t0):6.3f}")
