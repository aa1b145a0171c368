#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl; mkdir -p gpurun_out/tl
rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/tl -o p --output-format csv -- python bench.py ${WL:+--workload $WL --quick} --steps 30 --warmup 5 --prewarm 30 --no-cpu-baseline > gpurun_out/tl/log.txt 2>&1
python - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob('gpurun_out/tl/*kernel_trace.csv')[0])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:44], r.get('Queue_Id', '')) for r in rows]
try:
    mc = list(csv.DictReader(open(glob.glob('gpurun_out/tl/*memory_copy_trace.csv')[0])))
    ev += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r['Direction'][:20], '') for r in mc]
except Exception as e:
    print('no copy trace', e)
ev.sort()
# one step = from a k0_pack to the next k0_pack; take the last complete step
starts = [i for i, e in enumerate(ev) if e[2].startswith('k0_pack')]
import os; k = int(os.environ.get("STEP", "-3")); a, b = starts[k], starts[k + 1]
t0 = ev[a][0]
for s, e, n, q in ev[a:b]:
    print("%8.1f %8.1f %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
print("step span us", (ev[b][0] - t0) / 1e3)
PY
python - <<'PY'
# GPU idle time inside the printed step: intervals in which no kernel / copy of any queue runs
import csv, glob
rows = list(csv.DictReader(open(glob.glob('gpurun_out/tl/*kernel_trace.csv')[0])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:44]) for r in rows]
try:
    ev += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY') for r in csv.DictReader(open(glob.glob('gpurun_out/tl/*memory_copy_trace.csv')[0]))]
except Exception: pass
ev.sort()
starts = [i for i, e in enumerate(ev) if e[2].startswith('k0_pack')]
import os
k = int(os.environ.get('STEP', '-3'))
a, b = starts[k], starts[k + 1]
cur = ev[a][0]; idle = 0; gaps = []
for s, e, n in ev[a:b]:
    if s > cur: idle += s - cur; gaps.append(((s - cur) / 1e3, n))
    cur = max(cur, e)
print("idle us %.1f of %.1f; gaps > 8 us:" % (idle / 1e3, (ev[b][0] - ev[a][0]) / 1e3), ["%.0f before %s" % g for g in gaps if g[0] > 8])
PY
