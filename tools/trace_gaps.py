#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV (the host not keeping the queue fed, or
a dependency bubble): prints the busy / idle split of the traced interval and the largest gaps with the kernels on
either side.  usage: trace_gaps.py TRACE_DIR [skip_first_n_kernels]"""
import csv
import glob
import re
import sys

d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
f = sorted(glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[skip:]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n).split("(")[0][:60]


busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
gaps = []
end = int(rows[0]["End_Timestamp"])
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - end
    if g > 0:
        gaps.append((g, short(a["Kernel_Name"]), short(b["Kernel_Name"])))
    end = max(end, int(b["End_Timestamp"]))
idle = sum(g for g, _, _ in gaps)
print(f"{len(rows)} kernels, span {span / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms, idle {idle / 1e6:.3f} ms "
      f"({100.0 * idle / span:.1f} %)")
by_pair = {}
for g, a, b in gaps:
    k = (a, b)
    c = by_pair.setdefault(k, [0, 0])
    c[0] += 1
    c[1] += g
print("largest idle totals by (kernel before -> kernel after):")
for (a, b), (n, t) in sorted(by_pair.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {t / 1e3:9.1f} us in {n:4d} gaps ({t / n / 1e3:6.1f} us each)  {a}  ->  {b}")
