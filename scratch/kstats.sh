#!/bin/bash
# per-kernel average durations of the eager decode workload (profiles/pmc_decode.py) from a rocprofv3 kernel trace; prints the dec_* rows
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kst && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -- python3 $R/profiles/pmc_decode.py > /tmp/kst.log 2>&1 || { tail -5 /tmp/kst.log; exit 1; }
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/kst/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0][-52:]
    if "dec_" in n or "argmax" in n:
        print(f"{n:55s} calls {r['Calls']:>6s}  avg {float(r['AverageNs'])/1000:8.2f} us  min {float(r['MinNs'])/1000:8.2f}  max {float(r['MaxNs'])/1000:8.2f}")
PY
