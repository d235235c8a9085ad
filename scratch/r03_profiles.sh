#!/bin/bash
# round-3 profile set, one box: (1) rocprofv3 kernel trace + stats of the bench command, (2) FETCH_SIZE pass of the eager decode workload, (3) SQ counter passes,
# summaries written under gpurun_out/r03/ (the raw rocprof directories stay on the box)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
rm -rf $O && mkdir -p $O
cd /tmp
CMD="python3 $R/bench.py --steps 64 --warmup 8 --no-cpu-baseline --vit-batch 2"
rm -rf /tmp/p1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- $CMD > $O/bench_under_rocprof.log 2>&1
python3 $R/profiles/summarize.py /tmp/p1 $O/r03_bench_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline --vit-batch 2"
cp $(ls /tmp/p1/*/*kernel_stats.csv | head -1) $O/r03_bench_kernel_stats.csv
echo "stats done"
rm -rf /tmp/p2 && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p2 -- python3 $R/profiles/pmc_decode.py > $O/pmc_fetch.log 2>&1
python3 $R/profiles/pmc_summarize.py /tmp/p2 $O/r03_pmc_traffic.json > $O/r03_pmc_fetch_size_table.md
echo "fetch done"
cd $R && python3 bench.py > $O/r03_bench_line.json 2> $O/bench.err
echo "bench done"; tail -c 600 $O/r03_bench_line.json
