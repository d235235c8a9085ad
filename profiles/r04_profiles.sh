#!/bin/bash
# round-4 profile set, one box: (1) rocprofv3 kernel trace + stats of the bench command, (2) FETCH_SIZE pass of the eager decode workload, (3) MFMA-busy pass of the prefill
# workload, (4) SQ counter passes of the decode kernels, (5) the bench line itself (which reads the JSONs of 2 and 3).  Summaries under gpurun_out/r04/; the raw rocprof
# directories stay on the box.  Counter passes carry --kernel-trace only (no other trace domain beside --pmc).
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
rm -rf $O && mkdir -p $O
cd /tmp
CMD="python3 $R/bench.py --steps 64 --warmup 8 --no-cpu-baseline --vit-batch 2 --no-batched"
rm -rf /tmp/p1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- $CMD > $O/bench_under_rocprof.log 2>&1
python3 $R/profiles/summarize.py /tmp/p1 $O/r04_bench_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline --vit-batch 2 --no-batched"
cp $(ls /tmp/p1/*/*kernel_stats.csv | head -1) $O/r04_bench_kernel_stats.csv
echo "stats done"
rm -rf /tmp/p2 && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p2 -- python3 $R/profiles/pmc_decode.py > $O/pmc_fetch.log 2>&1
python3 $R/profiles/pmc_summarize.py /tmp/p2 $O/r04_pmc_traffic.json > $O/r04_pmc_fetch_size.md
echo "fetch done"
rm -rf /tmp/p3 && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/p3 -- python3 $R/profiles/pmc_prefill.py > $O/pmc_mfma.log 2>&1
python3 $R/profiles/pmc_mfma_summarize.py /tmp/p3 $O/r04_pmc_mfma.json > $O/r04_pmc_mfma.md
echo "mfma done"; cat $O/r04_pmc_mfma.md
for P in "a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "b SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD" "c GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT"; do
  set -- $P; t=$1; shift
  rm -rf /tmp/p4$t && rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/p4$t -- python3 $R/profiles/pmc_decode.py > $O/pmc_sq_$t.log 2>&1
done
python3 - <<'PY'
import collections, csv, glob, os
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for t in "abc":
    for f in glob.glob(f"/tmp/p4{t}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-48:]
            if "dec_" in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{R}/gpurun_out/r04/r04_pmc_sq.md", "w") as o:
    o.write("# rocprofv3 --pmc <SQ counters> --kernel-trace -- python3 profiles/pmc_decode.py (round 4, three passes; median per kernel over the decode launches of the final library)\n\n")
    names = sorted({c for k in agg for c in agg[k]})
    o.write("| kernel | " + " | ".join(names) + " |\n|---|" + "---|" * len(names) + "\n")
    for k in sorted(agg):
        o.write(f"| `{k}` | " + " | ".join(f"{sorted(agg[k][c])[len(agg[k][c]) // 2]:.0f}" if agg[k][c] else "-" for c in names) + " |\n")
print(open(f"{R}/gpurun_out/r04/r04_pmc_sq.md").read())
PY
echo "sq done"
cp $O/r04_pmc_traffic.json $O/r04_pmc_mfma.json $R/profiles/ 2>/dev/null || true
cd $R && python3 bench.py > $O/r04_bench_line.json 2> $O/bench.err
echo "bench done"; tail -c 700 $O/r04_bench_line.json
