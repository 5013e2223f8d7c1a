"""Idle gaps on the GPU timeline of a rocprofv3 --kernel-trace CSV: merges the kernel intervals of all streams and lists
where the device sat idle (gap, the kernel before and the kernel after).  usage: python tools/trace_gaps.py trace.csv [t0_frac t1_frac]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows))
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
T0, T1 = ks[0][0], ks[-1][1]
a, b = T0 + lo * (T1 - T0), T0 + hi * (T1 - T0)
ks = [k for k in ks if a <= k[0] <= b]
busy, idle, cur_end, prev = 0, 0, ks[0][0], ks[0][2]
gaps = []
for s, e, n in ks:
    if s > cur_end:
        idle += s - cur_end
        gaps.append((s - cur_end, prev, n))
        cur_end = s
    if e > cur_end:
        busy += e - cur_end
        cur_end = e
        prev = n
print("window %.1f ms: busy %.1f ms, idle %.1f ms (%d kernels)" % ((ks[-1][1] - ks[0][0]) / 1e6, busy / 1e6, idle / 1e6, len(ks)))
import collections
by = collections.defaultdict(lambda: [0, 0])
for g, p, n in gaps:
    by[(p, n)][0] += g; by[(p, n)][1] += 1
for (p, n), (g, c) in sorted(by.items(), key=lambda x: -x[1][0])[:14]:
    print("  %8.2f ms in %4d gaps (mean %6.1f us)  after %-50s before %s" % (g / 1e6, c, g / c / 1e3, p[:50], n[:50]))
