"""From a rocprofv3 --kernel-trace csv of the overlapped benchmark: how much wall time has NO convolution kernel running (the
matrix pipe idle), which kernels run in that time, and how busy the chip is overall.
usage: exposed_time.py kernel_trace.csv [skip_fraction]"""
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * skip
rows = [r for r in rows if r[0] >= lo]
span = rows[-1][1] - rows[0][0]
def union(iv):
    iv = sorted(iv); out = []; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: out.append((cs, ce)); cs, ce = s, e
        else: ce = max(ce, e)
    out.append((cs, ce)); return out
conv = union([(s, e) for s, e, n in rows if "conv_" in n])
allk = union([(s, e) for s, e, n in rows])
tc, ta = sum(e - s for s, e in conv), sum(e - s for s, e in allk)
neps = sum(1 for _, _, n in rows if "d2m_loss" in n)
print("span %.1f ms (%d episodes: %.2f ms each); some kernel running %.1f %%; a convolution kernel running %.1f %%; idle %.2f ms/episode; "
      "busy without convolution %.2f ms/episode" % (span / 1e6, neps, span / 1e6 / max(neps, 1), 100 * ta / span, 100 * tc / span,
                                                   (span - ta) / 1e6 / max(neps, 1), (ta - tc) / 1e6 / max(neps, 1)))
# which kernels fill the time without a convolution running
import bisect
cs = [c[0] for c in conv]
agg = collections.defaultdict(float)
for s, e, n in rows:
    if "conv_" in n: continue
    # part of [s, e) not covered by conv intervals
    t, cur = 0, s
    i = max(0, bisect.bisect_right(cs, s) - 1)
    while cur < e and i < len(conv):
        a, b = conv[i]
        if b <= cur: i += 1; continue
        if a >= e: break
        if a > cur: t += a - cur
        cur = max(cur, b); i += 1
    if cur < e: t += e - cur
    agg[n.split("(")[0].replace("void ", "")[:60]] += t
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:18]:
    print("  %7.3f ms/episode (kernel time outside convolution cover)  %s" % (v / 1e6 / max(neps, 1), k))
