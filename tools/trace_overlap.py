#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 --kernel-trace CSV: how busy is the GPU and how much do the
streams overlap?  usage: trace_overlap.py <dir> [fraction_of_trace_to_skip_at_start]"""
import csv, glob, sys, collections
d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
stop = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0   # fraction of the trace where the window ends (process teardown is idle time of no interest)
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows]
ev.sort()
t0, t1 = ev[0][0], max(e[1] for e in ev)
cut = t0 + skip * (t1 - t0)
ev = [e for e in ev if e[0] >= cut and e[1] <= t0 + stop * (t1 - t0)]
t0, t1 = ev[0][0], max(e[1] for e in ev)
span = t1 - t0
tot = sum(e[1] - e[0] for e in ev)
# union and concurrency histogram
pts = []
for s, e, _, _ in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
lvl = 0; last = pts[0][0]; hist = collections.Counter()
for t, dlt in pts:
    hist[lvl] += t - last; last = t; lvl += dlt
print(f"{f}\nwindow {span/1e6:.2f} ms, {len(ev)} dispatches, sum of durations {tot/1e6:.2f} ms (x{tot/span:.2f})")
for k in sorted(hist):
    print(f"  {k} kernels in flight: {hist[k]/1e6:8.2f} ms  {100*hist[k]/span:5.1f} %")
byq = collections.defaultdict(int)
for s, e, _, q in ev: byq[q] += e - s
print("per queue busy:", {q: f"{v/span:.2f}" for q, v in byq.items()})
# gaps between consecutive dispatches on the same queue
gaps = collections.defaultdict(list); prev = {}
for s, e, n, q in ev:
    if q in prev: gaps[q].append(s - prev[q])
    prev[q] = e
for q, g in gaps.items():
    g.sort(); print(f"queue {q}: median gap {g[len(g)//2]/1e3:.2f} us, mean {sum(g)/len(g)/1e3:.2f} us, p90 {g[int(.9*len(g))]/1e3:.2f} us")
