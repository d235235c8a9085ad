#!/usr/bin/env python3
"""Per-kernel FETCH_SIZE from a rocprofv3 --pmc run: python profiles/pmc_summarize.py <dir> [out.json] -> markdown on stdout (+ the JSON bench.py reads).
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B requests at 64 B -> x2; unit KiB.
The JSON carries, per fused decode kernel, the median HBM-side bytes per launch and, as `whole_token`, the sum over every `dec_*` dispatch divided by
the number of decode steps profiles/pmc_decode.py runs (24)."""
import collections, csv, glob, json, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0])))
agg = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] == "FETCH_SIZE":
        agg[r["Kernel_Name"].split("(")[0][-64:]].append(float(r["Counter_Value"]))
print("| kernel | dispatches | FETCH_SIZE median (KiB, raw) | HBM-side bytes / launch (x2 corrected) |\n|---|---|---|---|")
out = {"source": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 profiles/pmc_decode.py (own pass; median per kernel; x2 gfx950 correction)"}
keys = {"dec_gateup": "dec_gateup", "dec_attn": "dec_attn", "dec_proj_blk": "dec_down", "dec_qkv": "dec_qkv", "dec_proj_kernel": "dec_oproj", "dec_head": "dec_head"}
tok_total = 0.0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    vs = sorted(v); med = vs[len(vs) // 2]
    print(f"| `{k}` | {len(v)} | {med:.1f} | {med * 1024 * 2 / 1e6:.3f} MB |")
    if "dec_" in k:
        tok_total += sum(v) * 1024 * 2
    for pat, name in keys.items():
        if pat in k and name not in out:
            out[name] = {"fetch_bytes_per_launch": int(med * 1024 * 2), "dispatches": len(v), "kernel": k}
steps = 24
out["whole_token"] = {"fetch_bytes_per_token": int(tok_total / steps), "decode_steps": steps}
print(f"\nwhole decode token: {tok_total / steps / 1e6:.1f} MB fetched per token over {steps} steps")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
