"""Which copies a one-stream sweep makes: durations of the runtime's copy / fill kernels from a rocprofv3 --kernel-trace CSV, with the kernel
that ran before and after each.  usage: copy_hist.py <dir>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 3:]  # past the first sweep
def short(n):
    return n.replace("void ", "").replace("smoqy::", "").replace("(anonymous namespace)::", "").split("(")[0][:36]
groups = collections.defaultdict(list)
for i, r in enumerate(rows):
    if "copyBuffer" in r["Kernel_Name"] or "fillBuffer" in r["Kernel_Name"]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        prev = short(rows[i - 1]["Kernel_Name"]) if i else "-"
        nxt = short(rows[i + 1]["Kernel_Name"]) if i + 1 < len(rows) else "-"
        groups[(short(r["Kernel_Name"]), prev, nxt)].append(d)
tot = sum(sum(v) for v in groups.values())
print(f"copy / fill kernels in the window: {sum(len(v) for v in groups.values())} launches, {tot / 1e3:.2f} ms")
for k, v in sorted(groups.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print(f"{k[1]:>36s} -> {k[0]:24s} -> {k[2]:36s} n={len(v):4d} total {sum(v) / 1e3:7.3f} ms  avg {sum(v) / len(v):7.1f} us  max {max(v):7.1f}")
