#!/usr/bin/env python3
"""GPU busy fraction and the idle gaps of a rocprofv3 --kernel-trace run (CSV): tools/trace_gaps.py <dir> [skip_frac]
Takes the last (1 - skip_frac) of the trace (the timed steps), merges overlapping kernel intervals of all
streams and lists the largest gaps with the kernels on either side."""
import csv
import glob
import os
import sys

d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t0 + (t1 - t0) * skip
rows = [r for r in rows if r[0] >= cut]
busy, gaps = 0, []
cur_s, cur_e, last_name = rows[0][0], rows[0][1], rows[0][2]
for s, e, name in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name, name))
        cur_s, cur_e = s, e
        last_name = name
    elif e > cur_e:
        cur_e = e
        last_name = name
busy += cur_e - cur_s
span = cur_e - rows[0][0]
print("span %.3f ms  busy %.3f ms (%.1f %%)  kernels %d  gaps %d" % (span / 1e6, busy / 1e6, 100.0 * busy / span, len(rows), len(gaps)))
gaps.sort(reverse=True)
tot = sum(g[0] for g in gaps)
print("idle total %.3f ms; largest gaps:" % (tot / 1e6))
for g, a, b in gaps[:25]:
    print("  %8.1f us  after %-45s before %s" % (g / 1e3, a, b))
# gap histogram by preceding kernel
agg = {}
for g, a, b in gaps:
    k = agg.setdefault(a, [0, 0])
    k[0] += g
    k[1] += 1
print("idle by preceding kernel:")
for a, (g, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:15]:
    print("  %8.1f us  x%-4d %s" % (g / 1e3, n, a))
