#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV directory into a small text table."""
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {f}\n# total kernel time {tot/1e6:.2f} ms")
print(f"{'kernel':95s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>9s} {'pct':>6s}")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{r['Name'][:95]:95s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.2f} {float(r['AverageNs'])/1e3:9.2f} {float(r['MinNs'])/1e3:8.2f} {float(r['MaxNs'])/1e3:9.2f} {float(r['Percentage']):6.2f}")
