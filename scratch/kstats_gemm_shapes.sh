#!/bin/bash
# gemm_q4k_kernel launches of three prefills grouped by grid size (= shape): calls per prefill, average and total time
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kst && rocprofv3 --kernel-trace --output-format csv -d /tmp/kst -- python3 $R/scratch/prefill_once.py > /tmp/kst.log 2>&1 || { tail -5 /tmp/kst.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/kst/*/*kernel_trace.csv")[0]
g = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "gemm_q4k_kernel" in r["Kernel_Name"]:
        key = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])), int(r["Grid_Size_Z"]))
        g[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    v2 = v[len(v) // 3:]      # drop the first prefill (cold)
    print("grid %-18s calls/prefill %5.1f  avg %7.2f us  per prefill %8.1f us" % (k, len(v) / 3, sum(v2) / len(v2), sum(v2) / 2))
PY
