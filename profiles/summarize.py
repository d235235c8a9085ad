#!/usr/bin/env python3
"""Turns a rocprofv3 --kernel-trace --stats output directory into the committed summary (markdown):
   python profiles/summarize.py gpurun_out/prof4 profiles/r01_bench_kernel_stats.md "command line" """
import collections
import csv
import glob
import sys

d, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
stats = list(csv.DictReader(open(glob.glob(d + "/*/*kernel_stats.csv")[0])))
trace = list(csv.DictReader(open(glob.glob(d + "/*/*kernel_trace.csv")[0])))
trace.sort(key=lambda r: int(r["Start_Timestamp"]))
L = ["# rocprofv3 --kernel-trace --stats summary", "", f"command: `{cmd}`", "", "## per-kernel totals (whole run)", "",
     "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
tot = sum(float(r["TotalDurationNs"]) for r in stats)
for r in stats[:28]:
    L.append("| `%s` | %s | %.3f | %.3f | %.3f | %.3f | %.1f |" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                                                  float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
nx = [i for i, r in enumerate(trace) if "dec_next" in r["Kernel_Name"]]
if len(nx) > 40:
    # the timed graph replays: from the 20th dec_next on, for as long as a token keeps the same number of launches (the bench's later legs -- kernels timed alone, the
    # launch-by-launch steps -- also end in dec_next or sit between two of them and are left out)
    per = [nx[i + 1] - nx[i] for i in range(len(nx) - 1)]
    mode = collections.Counter(per[20:]).most_common(1)[0][0]
    last = 20
    while last < len(per) and per[last] == mode: last += 1
    dec = trace[nx[20] + 1:nx[last] + 1]
    ntok = last - 20
    t, c = collections.Counter(), collections.Counter()
    for r in dec:
        n = r["Kernel_Name"].split("(")[0][-60:]
        t[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        c[n] += 1
    L += ["", f"## decode steady state (tokens {ntok}; launches/token {len(dec) / ntok:.0f}; sum of kernel time/token {sum(t.values()) / ntok / 1e3:.1f} us)", "",
          "| kernel | calls/token | avg us | us/token |", "|---|---|---|---|"]
    for n, v in t.most_common():
        L.append("| `%s` | %.1f | %.2f | %.1f |" % (n, c[n] / ntok, v / c[n] / 1e3, v / ntok / 1e3))
open(out, "w").write("\n".join(L) + "\n")
print("\n".join(L[:12]))
