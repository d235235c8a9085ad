#!/bin/bash
# where do the gate|up rows come from after the attention launch has warmed them?  TCC hit / miss per dispatch, eager and graph-replayed decode
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_l2; rm -rf $O; mkdir -p $O
cat > /tmp/dec_graph.py <<PY
import os, sys
sys.path.insert(0, "$R")
import numpy as np
from mllm_amd import lib, synth
from tests.fixtures import weights
cfg = synth.qwen2vl_2b()
m = lib.Qwen2VL(cfg, weights.qwen2vl_file(cfg, cache_dir=os.environ.get("MLLM_AMD_CACHE", "/tmp/mllm_amd_cache")))
ids = (np.arange(40) * 7919 % 150000).astype(np.int32)
tok, _, _ = m.prefill(ids, want_logits=False)
gen, _ = m.generate(tok, 24)
print("tokens", gen[:8].tolist())
m.close()
PY
for mode in eager graph; do
  if [ $mode = eager ]; then export MLLM_HIP_NO_GRAPH=1; else unset MLLM_HIP_NO_GRAPH; fi
  rm -rf /tmp/pl2 && rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d /tmp/pl2 -- python3 /tmp/dec_graph.py > $O/$mode.log 2>&1 || { tail -5 $O/$mode.log; }
  python3 - $mode <<'PY'
import collections, csv, glob, sys
fs = glob.glob("/tmp/pl2/*/*counter_collection.csv")
if not fs: print(sys.argv[1], "no counters"); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0][-44:]
    if "dec_" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("mode", sys.argv[1])
for k in sorted(agg):
    print(f"  {k:46s}", "  ".join(f"{c} {sorted(v)[len(v)//2]:.0f} (n={len(v)})" for c, v in sorted(agg[k].items())))
PY
done
