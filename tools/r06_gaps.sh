#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "smoke or demo or strand or edge or two_contig or pipelined or async or composition" 2>&1 | tail -2
for k in 1 2 3; do
  timeout 600 python bench.py --quick --steps 100 --warmup 10 2>/dev/null | tail -1 > $O/bench_g.json
  python - <<PY
import json
d=json.load(open("$O/bench_g.json"))
print("bench", d["value"], d["ms_per_step"], d["step_ms"]["p50"], d["step_ms"]["p99"], d["roofline"]["frac"])
PY
done
rm -rf $O/trace_nt
HT_TIMERS=0 rocprofv3 --kernel-trace -d $O/trace_nt -o t --output-format csv -- python tools/host_trace.py > $O/trace_nt.log 2>&1
python tools/timeline.py $O/trace_nt > $O/timeline_nt.txt
grep "median" $O/trace_nt.log
sed -n 2,60p $O/timeline_nt.txt
