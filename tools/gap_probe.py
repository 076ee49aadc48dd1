#!/usr/bin/env python3
"""Where does a single-walker CG iteration's time go?  From a rocprofv3 --kernel-trace CSV of tools/one_stream.py: per (previous kernel ->
next kernel) pair inside the CG loop, the median gap between the end of one dispatch and the start of the next, plus the per-iteration
budget (kernel durations + gaps).  usage: gap_probe.py <dir>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
ev = ev[len(ev) // 2:]  # steady state


def short(n):
    # template arguments differ from build to build ("tfft_kernel<2, false>", "cheb_own_kernel<3, true, 3>"): classify by the stem and, for
    # the tau-FFT, its mode
    for mode in "0123":
        if f"tfft_kernel<{mode}," in n:
            return f"tfft_kernel<{mode}>"
    for key in ("cheb_own", "cheb_fast", "cheb_generic", "fdm_stream", "fdm_own", "fdm_fast", "fdm_kernel", "lanczos", "copyBuffer", "fillBuffer"):
        if key in n:
            return key
    return n.replace("void ", "").replace("smoqy::", "").replace("(anonymous namespace)::", "").split("(")[0][-28:]


gaps, durs = collections.defaultdict(list), collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    gaps[(short(n0), short(n1))].append(s1 - e0)
    durs[short(n0)].append(e0 - s0)
print(f"{'pair':48s} {'count':>7s} {'median gap us':>14s} {'mean':>8s} {'p90':>8s}")
for k, g in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:14]:
    g.sort()
    print(f"{k[0] + ' -> ' + k[1]:48s} {len(g):7d} {g[len(g) // 2] / 1e3:14.2f} {sum(g) / len(g) / 1e3:8.2f} {g[int(.9 * len(g))] / 1e3:8.2f}")
print("\nwhere the idle time of the window is (pairs by summed gap):")
for k, g in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:16]:
    print(f"{k[0] + ' -> ' + k[1]:48s} {len(g):7d} {sum(g) / 1e6:11.2f} ms")
loop = ["fdm_stream" if durs.get("fdm_stream") else "fdm_own", "tfft_kernel<2>", "cheb_own", "tfft_kernel<3>"]
tot_k = sum(sorted(durs[k])[len(durs[k]) // 2] for k in loop if durs[k]) / 1e3
tot_g = sum(sorted(gaps[(a, b)])[len(gaps[(a, b)]) // 2] for a, b in zip(loop, loop[1:] + loop[:1]) if gaps[(a, b)]) / 1e3
print(f"one iteration (medians): kernels {tot_k:.1f} us + boundaries {tot_g:.1f} us = {tot_k + tot_g:.1f} us")
span = ev[-1][1] - ev[0][0]
busy = sum(e - s for s, e, _ in ev)
print(f"window {span / 1e6:.2f} ms, kernels busy {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %)")
print(f"{len(ev)} dispatches in the window: one every {span / len(ev) / 1e3:.1f} us against an average kernel of {busy / len(ev) / 1e3:.1f} us — under the tracer the host's launch path "
      "is slower than without it, and where the first figure is the larger one the stream waits for the HOST, not for a poll")
it_busy = sum(sum(durs[k]) for k in loop if durs[k])
print(f"of which the four kernels of the CG iteration {it_busy / 1e6:.2f} ms ({100 * it_busy / span:.1f} % of the window); other kernels {(busy - it_busy) / 1e6:.2f} ms")
# the tail of the iteration-closing boundary: how many of its gaps are not back to back, and what they add up to
g = sorted(gaps[(loop[3], loop[0])])
big = [x for x in g if x > 2000]
print(f"{loop[3]} -> {loop[0]}: {len(g)} boundaries, {len(big)} above 2 us adding up to {sum(big) / 1e6:.2f} ms (largest {max(g) / 1e3:.1f} us)" if g else "")
