#!/usr/bin/env python3
"""Per-kernel FETCH_SIZE from a rocprofv3 --pmc run: python profiles/pmc_summarize.py <dir> -> markdown on stdout.
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B requests at 64 B -> x2; unit KiB."""
import collections, csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0])))
agg = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] == "FETCH_SIZE":
        agg[r["Kernel_Name"].split("(")[0][-64:]].append(float(r["Counter_Value"]))
print("| kernel | dispatches | FETCH_SIZE median (KiB, raw) | HBM-side bytes / launch (x2 corrected) |\n|---|---|---|---|")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v.sort(); med = v[len(v) // 2]
    print(f"| `{k}` | {len(v)} | {med:.1f} | {med * 1024 * 2 / 1e6:.3f} MB |")
