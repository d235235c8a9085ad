#!/bin/bash
# the reference's 2B Module through the adapter: the lazy window's fused launches against one launch per Op (MLLM_HIP_NO_FUSE=1), then the kernel trace of the fused form
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
for i in 1 2; do
  $R/oracle/_ref/ref_hip_qwen2vl --model $P --ids /tmp/ad/ids.i32 --steps 129 --threads 4 --out /tmp/ad --cfg $C --dump-every 0 2>&1 | grep -E "backend|Decoding" | cut -c1-260 | sed 's/^/fused:    /'
  cp /tmp/ad/tokens.i32 /tmp/ad/tokens_fused.i32
  MLLM_HIP_NO_FUSE=1 $R/oracle/_ref/ref_hip_qwen2vl --model $P --ids /tmp/ad/ids.i32 --steps 129 --threads 4 --out /tmp/ad --cfg $C --dump-every 0 2>&1 | grep -E "backend|Decoding" | cut -c1-260 | sed 's/^/per-op:   /'
  cmp /tmp/ad/tokens.i32 /tmp/ad/tokens_fused.i32 && echo "token ids equal"
done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pa
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -- $R/oracle/_ref/ref_hip_qwen2vl --model $P --ids /tmp/ad/ids.i32 --steps 33 --threads 4 --out /tmp/ad --cfg $C --dump-every 0 > /tmp/ad/log 2>&1
grep backend /tmp/ad/log | cut -c1-300
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/pa/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print("kernel time total ms", tot / 1e6, "calls", calls)
for r in rows[:14]:
    print(r["Name"].split("(")[0][-56:], r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 2), round(float(r["AverageNs"]) / 1e3, 2))
PY
