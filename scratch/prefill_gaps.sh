#!/bin/bash
# busy time vs span of one image prefill of the 2 B engine (kernel trace): how much of the 13.6 ms is gaps between kernels
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pg
rocprofv3 --kernel-trace --output-format csv -d /tmp/pg -- python3 $R/profiles/pmc_prefill.py > /tmp/pg.log 2>&1
tail -2 /tmp/pg.log
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pg/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last prefill: kernels after the last big gap (> 200 us)
starts = [int(r["Start_Timestamp"]) for r in rows]; ends = [int(r["End_Timestamp"]) for r in rows]
cut = 0
for i in range(1, len(rows)):
    if starts[i] - ends[i - 1] > 200000: cut = i
seg = rows[cut:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
gaps = [int(seg[i]["Start_Timestamp"]) - int(seg[i - 1]["End_Timestamp"]) for i in range(1, len(seg))]
print("kernels", len(seg), "busy ms", busy / 1e6, "span ms", span / 1e6, "gap total ms", sum(g for g in gaps if g > 0) / 1e6, "mean gap us", sum(gaps) / len(gaps) / 1e3)
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = r["Kernel_Name"].split("(")[0][-44:]; agg[k][0] += 1; agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]: print(f"{k:46s} {n:5d} {t / 1e6:8.3f} ms {t / n / 1e3:8.2f} us")
big = sorted(gaps, reverse=True)[:8]
print("largest gaps us", [round(g / 1e3, 1) for g in big])
PY
