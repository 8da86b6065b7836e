"""Reads a rocprofv3 kernel-trace CSV of the (overlapped, three-stream) benchmark and reports, for the last `--episodes` episodes'
worth of time: GPU busy time (union of kernel intervals), sum of kernel durations, and the per-kernel totals.
usage: python tools/overlap_trace.py <r_kernel_trace.csv> [t_skip_fraction [t_end_fraction]]"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * skip
t_hi = rows[0][0] + (rows[-1][1] - rows[0][0]) * (float(sys.argv[3]) if len(sys.argv) > 3 else 1.0)
rows = [r for r in rows if t_lo <= r[0] <= t_hi]
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
for s, e, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in rows)
print("span %.2f ms, GPU busy (union) %.2f ms = %.1f %%, sum of kernel durations %.2f ms (overlap factor %.2f)" % (span / 1e6, busy / 1e6, 100.0 * busy / span, tot / 1e6, tot / busy))
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in rows:
    k = n.split("(")[0][:90]
    agg[k][0] += e - s
    agg[k][1] += 1
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    print("%7.2f %%  %8.2f ms  %6d  %7.1f us  %s" % (100.0 * t / tot, t / 1e6, c, t / c / 1e3, k))
