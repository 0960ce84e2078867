#!/bin/bash
# tests/prof_launch_fixed.sh: duration of the camera-ray launch of k_rays for several shardings of the headline frame
# (one bounce only), from rocprofv3 kernel traces: what a launch costs beside its rays
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/launch_fixed
mkdir -p $OUT
run() { # tag, bench args...
    tag=$1; shift
    timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/$tag -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --max-bounces 1 "$@" > $OUT/$tag.json 2> $OUT/$tag.log || { echo "$tag failed"; tail -3 $OUT/$tag.log; return 1; }
    python3 - $OUT/$tag "$tag" <<'PY'
import csv, glob, sys
trace = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
ks = sorted(((r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Grid_Size_X']) for r in csv.DictReader(open(trace))), key=lambda x: x[1])
gi = [i for i, k in enumerate(ks) if 'generate' in k[0]]
seq = [k for k in ks[gi[-1]:] if 'rocclr' not in k[0]][:5]
print(f"{sys.argv[2]:28s} " + "  ".join(f"{k[0].split('(')[0].split('::')[-1][:14]} {(k[2]-k[1])/1e6:6.3f}" for k in seq))
PY
}
run whole
run shard_0of2 --shard 0/2
run shard_0of4 --shard 0/4
run shard_0of8 --shard 0/8
run shard_0of16 --shard 0/16
run shard_0of8_tile64 --shard 0/8 --tile-size 64
run shard_0of8_tile256 --shard 0/8 --tile-size 256
run shard_3of8 --shard 3/8
