#!/usr/bin/env python3
"""Timeline of one rocprofv3 --kernel-trace run: per kernel name count / mean duration, and the busy vs idle
share of the window between the first and the last stage-kernel launch (gaps = launch latency, stream waits,
transfers).  usage: trace_gaps.py <dir with *_kernel_trace.csv> [skip_first_n_dispatches]"""
import csv
import glob
import sys
import collections

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void pion::", "")[:60],
                     r.get("Stream_Id", "?")))
rows.sort()
st = [r for r in rows if "k_stage" in r[2]]
# steady state: last 40 % of the stage launches
t0 = st[int(len(st) * 0.6)][0]
rows = [r for r in rows if r[0] >= t0]
acc = collections.defaultdict(list)
for s, e, n, q in rows:
    acc[(n, q)].append((e - s) / 1e3)
for (n, q), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("%-62s stream %-4s n %4d  mean %9.1f us  total %9.1f us" % (n, q, len(v), sum(v) / len(v), sum(v)))
# union busy time
iv = sorted((s, e) for s, e, _, _ in rows)
busy = 0
cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
span = iv[-1][1] - iv[0][0]
print("window %.1f us, some kernel running %.1f us (%.1f %%), idle %.1f us" % (span / 1e3, busy / 1e3, 100.0 * busy / span, (span - busy) / 1e3))
