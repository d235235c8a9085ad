#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kst && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -- python3 $R/scratch/prefill_once.py > /tmp/kst.log 2>&1 || { tail -5 /tmp/kst.log; exit 1; }
grep "prefill ms" /tmp/kst.log
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/kst/*/*kernel_stats.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
tot = 0
for r in rows:
    n = r["Name"].split("(")[0][-60:]
    t = float(r["TotalDurationNs"]) / 3000
    tot += t
    if t > 20: print(f"{n:62s} calls/prefill {int(r['Calls'])/3:7.1f}  avg {float(r['AverageNs'])/1000:8.2f} us  per prefill {t:9.1f} us")
print("sum per prefill", tot, "us")
PY
