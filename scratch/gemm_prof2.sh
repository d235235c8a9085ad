#!/bin/bash
# kernel-only GEMM times per shape: rocprofv3 kernel trace of scratch/gemm_time2.py; usage: scratch/gemm_prof.sh <so> <tag>
export TMPDIR=/tmp
rm -rf gpurun_out/gp_$2
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gp_$2 -- python3 scratch/gemm_time2.py $1 > gpurun_out/gp_$2.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob('gpurun_out/gp_$2/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'gemm_q4k' in r['Kernel_Name']]
shapes = ((1024,5120,1280),(1024,5120,2560),(1024,5120,5120),(1024,2560,1280),(512,5120,1280),(1024,1024,1280))
out = []
for i, s in enumerate(shapes):
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000 for r in rows[i*23+3:(i+1)*23]]
    out.append('%dx%dx%d %.1f' % (s + (sum(d) / len(d),)))
print('$2:', ' | '.join(out), '| vgpr', rows[0]['VGPR_Count'])
PY
