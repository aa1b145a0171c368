#!/bin/bash
# round 6, first GPU call: the pileup stage's ceiling kernels, and the hunt for the 60-85 ms stalls of fresh contexts
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
python tools/ceiling.py c3 > $O/ceiling_c3.txt 2> $O/ceiling_c3.err
python tools/stall_hunt.py --passes 40 > $O/stall_hunt.jsonl 2> $O/stall_hunt.err
rocprofv3 --hip-trace --kernel-trace -d $O/stall_trace -o t --output-format csv -- python tools/stall_hunt.py --passes 24 --variants bench > $O/stall_hunt_traced.jsonl 2> $O/stall_trace.log
python - $O <<'PY' > $O/stall_trace_long_calls.txt
import csv, glob, sys
O = sys.argv[1]
f = glob.glob(O + "/stall_trace/**/*hip_api_trace.csv", recursive=True)
print("files", f)
rows = []
for fn in f:
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Thread_Id", "")))
rows.sort()
print("hip api calls:", len(rows))
long_ = [i for i, r in enumerate(rows) if r[1] - r[0] > 5_000_000]
print("calls longer than 5 ms:", len(long_))
for i in long_[:80]:
    r = rows[i]
    print("%.3f ms  %s  thread %s   at t = %.3f s;  before it: %s" % ((r[1] - r[0]) / 1e6, r[2], r[3], (r[0] - rows[0][0]) / 1e9, " <- ".join(x[2] for x in rows[max(0, i - 4):i][::-1])))
import collections
agg = collections.defaultdict(lambda: [0, 0, 0])
for r in rows:
    a = agg[r[2]]; a[0] += 1; a[1] += r[1] - r[0]; a[2] = max(a[2], r[1] - r[0])
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-40s n %7d  total %9.2f ms  max %8.3f ms" % (k, a[0], a[1] / 1e6, a[2] / 1e6))
PY
rm -rf $O/stall_trace
tail -5 $O/ceiling_c3.err $O/stall_hunt.err | cut -c1-300
cat $O/ceiling_c3.txt
cut -c1-1500 $O/stall_hunt.jsonl
head -60 $O/stall_trace_long_calls.txt | cut -c1-260
