#!/bin/bash
# kernel trace of the reference's 2B Module through the adapter (lazy window on): per kernel AND per grid size (the three row_fused launches of a layer differ in grid)
R=$GRAFT_REPO_ROOT
cd $R
python3 - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, '.')
from mllm_amd import synth
from mllm_amd import synthfile as weights
from tests.test_gpu_adapter import _cfg_string
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
ids = (np.arange(24) * 7919 % 150000).astype(np.int32)
os.makedirs('/tmp/ad', exist_ok=True); ids.tofile('/tmp/ad/ids.i32')
open('/tmp/ad/cmd', 'w').write(f"{path}\n{_cfg_string(cfg)}\n")
PY
P=$(sed -n 1p /tmp/ad/cmd); C=$(sed -n 2p /tmp/ad/cmd)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pa
rocprofv3 --kernel-trace --output-format csv -d /tmp/pa -- $R/oracle/_ref/ref_hip_qwen2vl --model $P --ids /tmp/ad/ids.i32 --steps 65 --threads 4 --out /tmp/ad --cfg $C --dump-every 0 > /tmp/ad/log 2>&1
grep backend /tmp/ad/log | cut -c1-300
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pa/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]      # steady-state decode
agg = collections.defaultdict(list)
for r in rows:
    agg[(r["Kernel_Name"].split("(")[0][-44:], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("kernels", len(rows), "busy ms", tot / 1e6, "span ms", span / 1e6)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(k, len(v), "avg us", round(sum(v) / len(v) / 1e3, 2), "total ms", round(sum(v) / 1e6, 2))
PY
