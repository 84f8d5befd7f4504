#!/usr/bin/env python3
"""aggregate a rocprofv3 kernel_trace.csv: top kernels by total time, restricted to the last `frac` of the trace (the timed
steps, not the warm-up / MIOpen find); elementwise / copy kernels are further split by launch size so that the tensor
behind each one can be identified."""
import collections
import csv
import sys

path = sys.argv[1]
seconds = float(sys.argv[2][:-1]) if len(sys.argv) > 2 and sys.argv[2].endswith("s") else None      # "0.86s": the last 0.86 seconds
frac = 0.5 if seconds is not None else (float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
if seconds is not None:
    cut = t1 - int(seconds * 1e9)
elif frac >= 1.0:
    # frac = K >= 1: the window is the last K forwards, located by the 24 global-attention launches each one issues
    marks = [int(r["Start_Timestamp"]) for r in rows if ("vit_attn_sp_kernel" in r["Kernel_Name"]) or
             ("vit_attn_split_kernel" in r["Kernel_Name"] and ("Li2ELi8E" in r["Kernel_Name"] or "2, 8," in r["Kernel_Name"])) or
             ("flash_attn_kernel" in r["Kernel_Name"] and "Li80ELi2E" in r["Kernel_Name"])]
    k = int(frac) * 24
    cut = marks[-k] - 1500000 if len(marks) >= k else t0          # minus ~1.5 ms: the text encoder / patch embed before it
    print("window = last %d forwards" % int(frac))
else:
    cut = t1 - (t1 - t0) * frac
rows = [r for r in rows if int(r["Start_Timestamp"]) >= cut]
tot = collections.defaultdict(lambda: [0, 0])
split = collections.defaultdict(lambda: [0, 0])
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    k = r["Kernel_Name"][:110]
    tot[k][0] += d
    tot[k][1] += 1
    if "elementwise" in k or "copy" in k.lower() or "Cat" in k:
        threads = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * max(1, int(r.get("Grid_Size_Y", 1) or 1))
        key = (r["Kernel_Name"][:150], threads)
        split[key][0] += d
        split[key][1] += 1
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6
busy = sum(v[0] for v in tot.values()) / 1e6
print("window %.1f ms, GPU busy %.1f ms, %d dispatches" % (span, busy, len(rows)))
for k, (d, n) in sorted(tot.items(), key=lambda x: -x[1][0])[:45]:
    print("%9.2f ms %6d x %9.1f us  %s" % (d / 1e6, n, d / n / 1e3, k))
print("\nelementwise / copy kernels by launch size (threads):")
for (k, th), (d, n) in sorted(split.items(), key=lambda x: -x[1][0])[:30]:
    print("%9.2f ms %6d x %9.1f us  threads=%-10d %s" % (d / 1e6, n, d / n / 1e3, th, k))

# idle gaps inside the window: where the GPU waits for the host (a synchronisation or a launch-bound stretch)
gaps = []
prev_end, prev_name = None, None
for r in rows:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None and st > prev_end:
        gaps.append((st - prev_end, prev_name, r["Kernel_Name"][:70], (prev_end - cut) / 1e6))
    if prev_end is None or en > prev_end:
        prev_end, prev_name = en, r["Kernel_Name"][:70]
tot_gap = sum(g[0] for g in gaps)
# idle time per 50 ms slice of the window: where in the step the GPU starves
slices = collections.defaultdict(float)
for d, _, _, at in gaps:
    slices[int(at // 50)] += d / 1e6
print("\nidle ms per 50 ms slice of the window: " + " ".join("%d:%.0f" % (k * 50, v) for k, v in sorted(slices.items())))
print("\nidle gaps: total %.2f ms in %d gaps; > 20 us: %.2f ms in %d gaps" % (
    tot_gap / 1e6, len(gaps), sum(g[0] for g in gaps if g[0] > 20000) / 1e6, sum(1 for g in gaps if g[0] > 20000)))
for d, a, b, at in sorted(gaps, key=lambda x: -x[0])[:25]:
    print("%9.1f us  at +%7.1f ms  after %-70s before %s" % (d / 1e3, at, a, b))
by_next = collections.defaultdict(lambda: [0, 0])
for d, a, b, _ in gaps:
    by_next[b][0] += d
    by_next[b][1] += 1
print("\ngap time by the kernel that follows the gap:")
for k, (d, n) in sorted(by_next.items(), key=lambda x: -x[1][0])[:20]:
    print("%9.2f ms %6d x %7.1f us  %s" % (d / 1e6, n, d / n / 1e3, k))

print("\nkernels by launch count (per window):")
for k, (d, n) in sorted(tot.items(), key=lambda x: -x[1][1])[:60]:
    print("%6d x %9.1f us  %9.2f ms  %s" % (n, d / n / 1e3, d / 1e6, k))
