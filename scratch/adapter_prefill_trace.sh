#!/bin/bash
# kernel stats of the reference's 2B Module through the adapter: 448x448 image + 24 tokens prefill (S = 282), 3 decode steps
R=$GRAFT_REPO_ROOT
cd $R
python3 - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, '.')
from mllm_amd import synth
from mllm_amd import synthfile as weights
from tests.test_gpu_adapter import _cfg_string
cfg = synth.qwen2vl_2b(); path = weights.qwen2vl_file(cfg)
pix, grid, ids = synth.qwen2vl_inputs(cfg, (32, 32), 24)
os.makedirs('/tmp/ad', exist_ok=True); ids.astype(np.int32).tofile('/tmp/ad/ids.i32'); pix.astype(np.float32).tofile('/tmp/ad/pix.f32')
open('/tmp/ad/cmd', 'w').write(f"{path}\n{_cfg_string(cfg)}\n")
PY
P=$(sed -n 1p /tmp/ad/cmd); C=$(sed -n 2p /tmp/ad/cmd)
for i in 1 2; do $R/oracle/_ref/ref_hip_qwen2vl --model $P --ids /tmp/ad/ids.i32 --pix /tmp/ad/pix.f32 --grid 1,32,32 --steps 4 --threads 4 --out /tmp/ad --cfg $C --dump-every 0 2>&1 | grep -E "backend|TTFT" | cut -c1-200; done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pa
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -- $R/oracle/_ref/ref_hip_qwen2vl --model $P --ids /tmp/ad/ids.i32 --pix /tmp/ad/pix.f32 --grid 1,32,32 --steps 2 --threads 4 --out /tmp/ad --cfg $C --dump-every 0 > /tmp/ad/log 2>&1
grep backend /tmp/ad/log | cut -c1-200
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/pa/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print("kernel time total ms", tot / 1e6, "calls", calls)
for r in rows[:24]:
    print(r["Name"].split("(")[0][-56:], r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 2), round(float(r["AverageNs"]) / 1e3, 2))
PY
