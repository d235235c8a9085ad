#!/bin/bash
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
for opt in "merge_o=0 steps=16" "merge_o=0 steps=48" "merge_o=-1 steps=64"; do
  rm -rf /tmp/pm
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -- python3 $R/scratch/merge_prof.py $opt > /tmp/pm_run.log 2>&1
  grep "tok/s" /tmp/pm_run.log; head -40 /tmp/pm_run.log | cut -c1-220
  python3 - <<PY
import csv, glob
fs = glob.glob("/tmp/pm/**/*kernel_stats.csv", recursive=True)
print("$opt", fs[:1])
if fs:
    for r in csv.DictReader(open(fs[0])):
        n = r["Name"]
        if "dec_" in n:
            print("  ", n.split("(")[0][-52:], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2))
PY
done
