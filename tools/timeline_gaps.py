"""GPU busy/idle from a rocprofv3 --kernel-trace csv: union of kernel intervals, gaps, per-stream busy.
usage: timeline_gaps.py kernel_trace.csv [skip_fraction [end_fraction]]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "0"))) for r in rows))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
t0, t1 = ev[0][0], max(e[1] for e in ev)
lo = t0 + (t1 - t0) * skip
hi = t0 + (t1 - t0) * (float(sys.argv[3]) if len(sys.argv) > 3 else 1.0)
ev = [e for e in ev if lo <= e[0] <= hi]
span = max(e[1] for e in ev) - ev[0][0]
busy, cur_s, cur_e, gaps = 0, ev[0][0], ev[0][1], []
for s, e, n, q in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in ev)
print("span %.1f ms  busy(union) %.1f ms (%.1f%%)  sum of kernel durations %.1f ms  kernels %d" % (span / 1e6, busy / 1e6, 100.0 * busy / span, tot / 1e6, len(ev)))
g = sorted(gaps, reverse=True)
print("idle %.1f ms in %d gaps; >20us: %d gaps %.1f ms; top:" % ((span - busy) / 1e6, len(gaps), sum(1 for x in g if x[0] > 20000), sum(x[0] for x in g if x[0] > 20000) / 1e6))
for d, n in g[:12]: print("   %.0f us before %s" % (d / 1e3, n[:70]))
byq = collections.defaultdict(int)
for s, e, n, q in ev: byq[q] += e - s
print("per queue busy ms:", {k: round(v / 1e6, 1) for k, v in byq.items()})
