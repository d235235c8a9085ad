#!/bin/bash
# SQ counters of the decode kernels (own pmc passes, kernel trace only): how busy the VALU / LDS / memory side is per launch
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_sq && mkdir -p $R/gpurun_out/pmc_sq
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq/a -- python3 $R/profiles/pmc_decode.py > $R/gpurun_out/pmc_sq/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq/b -- python3 $R/profiles/pmc_decode.py > $R/gpurun_out/pmc_sq/b.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq/c -- python3 $R/profiles/pmc_decode.py > $R/gpurun_out/pmc_sq/c.log 2>&1
python3 - <<'PY'
import collections, csv, glob, os
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in "abc":
    for f in glob.glob(f"{R}/gpurun_out/pmc_sq/{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-48:]
            if "dec_" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{R}/gpurun_out/pmc_sq/summary.md", "w") as o:
    names = sorted({c for k in agg for c in agg[k]})
    o.write("| kernel | " + " | ".join(names) + " |\n|---|" + "---|" * len(names) + "\n")
    for k in sorted(agg):
        o.write(f"| `{k}` | " + " | ".join(f"{sorted(agg[k][c])[len(agg[k][c]) // 2]:.0f}" if agg[k][c] else "-" for c in names) + " |\n")
print(open(f"{R}/gpurun_out/pmc_sq/summary.md").read())
PY
find $R/gpurun_out/pmc_sq -name "*.csv" | head -20
for d in a b c; do f=$(find $R/gpurun_out/pmc_sq/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && head -3 $f; tail -3 $R/gpurun_out/pmc_sq/$d.log; done
rm -rf $R/gpurun_out/pmc_sq/a $R/gpurun_out/pmc_sq/b $R/gpurun_out/pmc_sq/c
