#!/usr/bin/env python3
"""top kernels of the LAST `seconds` of a rocprofv3 kernel_trace.csv (one steady-state step of a long run whose start is warm-up):
    python tools/last_window.py <kernel_trace.csv> <seconds> [rows]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
sec = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
t1 = max(int(r["End_Timestamp"]) for r in rows)
cut = t1 - int(sec * 1e9)
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
tot = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = r["Kernel_Name"][:120]
    tot[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot[k][1] += 1
busy = sum(v[0] for v in tot.values())
print("window %.3f s: %d dispatches, kernel time %.1f ms" % (sec, len(rows), busy / 1e6))
for k, (d, n) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:top]:
    print("%9.2f ms %6d  %s" % (d / 1e6, n, k))
