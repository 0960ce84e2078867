#!/bin/bash
# tests/prof_sq.sh OUTNAME [variant]: three SQ counter passes of one bench frame (RT_PIPELINE is inherited),
# or of the program PROF_CMD names
set -e
OUT=$PWD/gpurun_out/$1; V=${2:-base}
if [ "$V" != base ]; then export RT_LIB=$PWD/raytracer_amd/variants/librt_$V.so; fi
export TMPDIR=/tmp
B="${PROF_CMD:-python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline $BENCH_ARGS}"
mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/a -o p -- $B > $OUT/a.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD --output-format csv -d $OUT/b -o p -- $B > $OUT/b.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/c -o p -- $B > $OUT/c.log 2>&1
python3 - $OUT <<'PY'
import csv,glob,collections,sys,re
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob(sys.argv[1]+'/*/**/*counter_collection.csv',recursive=True)):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        m=re.search(r'(k_[a-z_]+)(<[a-z]+)?', n)
        if not m or 'Lb1E' in n or '<true' in n[:40]: continue
        agg[m.group(1)][r['Counter_Name']]+=float(r['Counter_Value'])
with open(sys.argv[1]+'/summary.txt','w') as o:
    for kern,c in sorted(agg.items()):
        if c.get('SQ_WAVE_CYCLES',0) < 1e6 and c.get('SQ_BUSY_CYCLES',0) < 1e6 and c.get('SQ_INSTS_SMEM',0) < 1e5: continue
        for k,v in sorted(c.items()):
            line=f"{kern:16s} {k:28s} {v:.6g}"
            print(line); o.write(line+"\n")
        if 'SQ_THREAD_CYCLES_VALU' in c and c.get('SQ_ACTIVE_INST_VALU'):
            line=f"{kern:16s} lanes_per_valu_inst           {c['SQ_THREAD_CYCLES_VALU']/c['SQ_ACTIVE_INST_VALU']:.2f}"
            print(line); o.write(line+"\n")
PY
