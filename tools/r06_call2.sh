#!/bin/bash
# round 6: the stall hunt over bench.py's literal seeds stage, untraced and under rocprofv3 --hip-trace
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
python tools/stall_hunt2.py 3 > $O/stall_hunt2.jsonl 2> $O/stall_hunt2.err
rocprofv3 --hip-trace --kernel-trace -d $O/stall_trace2 -o t --output-format csv -- python tools/stall_hunt2.py 3 > $O/stall_hunt2_traced.jsonl 2> $O/stall_trace2.log
python - $O <<'PY' > $O/stall_trace2_windows.txt
import csv, glob, sys, collections
O = sys.argv[1]
rows = []
for fn in glob.glob(O + "/stall_trace2/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Thread_Id", "")))
rows.sort()
print("hip api calls:", len(rows))
kern = []
for fn in glob.glob(O + "/stall_trace2/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
kern.sort()
win = []
t_open = None
for r in rows:
    if r[2] == "hipRuntimeGetVersion": t_open = r[1]
    elif r[2] == "hipDriverGetVersion" and t_open is not None: win.append((t_open, r[0])); t_open = None
print("timed windows:", len(win))
for wi, (a, b) in enumerate(win):
    inside = [r for r in rows if a <= r[0] <= b]
    longc = [r for r in inside if r[1] - r[0] > 2_000_000]
    kin = [k for k in kern if a <= k[0] <= b]
    # the largest gap between consecutive kernel starts inside the window (a stall of the host shows as an idle GPU)
    gap = max(((kin[i + 1][0] - kin[i][1]) / 1e6, kin[i][2], kin[i + 1][2]) for i in range(len(kin) - 1)) if len(kin) > 1 else (0, "", "")
    longk = sorted(((k[1] - k[0]) / 1e6, k[2]) for k in kin)[-2:]
    print("window %2d: %.2f ms, %d api calls, %d kernels; largest idle gap between kernels %.2f ms (%s -> %s); longest kernels %s" % (wi, (b - a) / 1e6, len(inside), len(kin), gap[0], gap[1], gap[2], longk))
    for r in longc:
        i = rows.index(r)
        print("    %.3f ms  %s  at +%.2f ms;  before it: %s" % ((r[1] - r[0]) / 1e6, r[2], (r[0] - a) / 1e6, " <- ".join(x[2] for x in rows[max(0, i - 5):i][::-1])))
PY
rm -rf $O/stall_trace2
cat $O/stall_hunt2.jsonl | cut -c1-600
echo ---- traced
cat $O/stall_hunt2_traced.jsonl | cut -c1-600
cat $O/stall_trace2_windows.txt | cut -c1-400
tail -n 5 $O/stall_hunt2.err
