#!/usr/bin/env python3
"""Matrix-pipe busy fraction per prefill kernel from a rocprofv3 `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace` run of profiles/pmc_prefill.py:
    python profiles/pmc_mfma_summarize.py <dir> [out.json]   -> markdown on stdout (+ the JSON bench.py reads as prefill_roofline.kernels[].mfma_busy_frac)
SQ_VALU_MFMA_BUSY_CYCLES sums the cycles the matrix pipes of all SIMDs are busy (MI355X_MICROARCH.md: = 32 x N_mfma for v_mfma_f32_32x32x16_*); busy fraction =
busy cycles / (kernel duration x 2.4 GHz x 1,024 SIMDs), durations from the same (counter-collecting) run's kernel trace."""
import collections, csv, glob, json, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(d + "/*/*counter_collection.csv")[0])))
busy = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        busy[r["Kernel_Name"].split("(")[0][-64:]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(d + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0][-64:]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("| kernel | dispatches | MFMA busy cycles / launch (mean) | mean us | MFMA busy fraction |\n|---|---|---|---|---|")
out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -- python3 profiles/pmc_prefill.py (own pass); busy / (duration x 2.4 GHz x 1024 SIMDs)"}
for k, v in sorted(busy.items(), key=lambda kv: -sum(kv[1])):
    if not dur.get(k) or sum(v) == 0:
        continue
    b = sum(v) / len(v)
    us = sum(dur[k]) / len(dur[k]) / 1e3
    frac = b / (us * 1e-6 * 2.4e9 * 1024)
    print(f"| `{k}` | {len(v)} | {b:.3e} | {us:.1f} | {100 * frac:.1f} % |")
    for pat, name in (("gemm_q4k_kernel", "gemm_q4k"), ("fa2_prefill_kernel<80", "fa2_prefill_80"), ("fa2_prefill_kernel<64", "fa2_prefill_64"), ("fa2_prefill_kernel<128", "fa2_prefill_128"),
                      ("gemm_f32_mfma", "gemm_f32_mfma")):
        if pat in k:
            e = out.setdefault(name, {"busy_cycles": 0.0, "us": 0.0, "dispatches": 0})
            e["busy_cycles"] += sum(v); e["us"] += sum(dur[k]) / 1e3 * (len(v) / len(dur[k])); e["dispatches"] += len(v)
for name, e in out.items():
    if isinstance(e, dict):
        e["mfma_busy_frac"] = round(e["busy_cycles"] / (e["us"] * 1e-6 * 2.4e9 * 1024), 4)
        e["mean_us"] = round(e["us"] / e["dispatches"], 2)
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
