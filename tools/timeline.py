"""Kernel timeline of two steady-state steps from a rocprofv3 --kernel-trace CSV: start, end, queue, kernel, duration and the gap to the previous
kernel of the same queue.  usage: timeline.py <dir with *kernel_trace.csv> [steps from the end, default 12]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 12
def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return re.split(r"\(", n)[0][:36]
K = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in csv.DictReader(open(f)))
marks = [i for i, k in enumerate(K) if k[2].startswith("k0_bind_a")]
m, end = marks[-back], K[marks[-back + 2]][0]
t0 = K[m][0]
last = {}
print("us from a k0_bind_a: start end | queue | kernel | duration | gap behind the queue's previous kernel")
for s, e, n, q in K[max(0, m - 6):]:
    if s >= end: break
    gap = (s - last[q]) / 1e3 if q in last else float("nan")
    last[q] = e
    print("%9.1f %9.1f  q%-3s %-36s %8.1f %8.1f" % ((s - t0) / 1e3, (e - t0) / 1e3, q, n, (e - s) / 1e3, gap))
