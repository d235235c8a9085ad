#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $R/scratch/topp_prof.py > /tmp/pp.log 2>&1 || { tail -5 /tmp/pp.log; exit 1; }
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/pp/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(r["Name"].split("(")[0][-60:], r["Calls"], round(float(r["TotalDurationNs"])/1e6,2), round(float(r["AverageNs"])/1e3,2))
PY
