"""From a rocprofv3 --kernel-trace csv of the overlapped benchmark: the kernels between the loss kernel and the first kernel of the
trunk's backward (the heads' backward) of one steady-state episode, per stream, with start offsets, durations and gaps.
usage: head_timeline.py kernel_trace.csv"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:70], r["Stream_Id"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
loss = [i for i, r in enumerate(rows) if "d2m_loss" in r[2]]
i0 = loss[len(loss) * 2 // 3]
t0 = rows[i0][0]
# forward side: from the last trunk kernel before the loss
j = i0
while j > 0 and "adaptive_maxpool_mean_kernel" not in rows[j][2]:
    j -= 1
print("--- heads forward (from the last pooling kernel of the trunk to the loss)")
for s, e, n, st in rows[j:i0 + 1]:
    print("%9.1f us  +%7.1f us  stream %-3s %s" % ((s - rows[j][0]) / 1e3, (e - s) / 1e3, st, n))
print("--- heads backward (loss -> first pooling-backward kernel of the trunk)")
k = i0
while k < len(rows) and "adaptive_maxpool_mean_bwd" not in rows[k][2]:
    k += 1
for s, e, n, st in rows[i0:k + 1]:
    print("%9.1f us  +%7.1f us  stream %-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, st, n))
