#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "smoke or demo or strand or edge or two_contig" 2>&1 | tail -2
for of in 1 0 1 0; do
  LCR_OWN_FILL=$of timeout 600 python bench.py --quick --steps 100 --warmup 10 2>/dev/null | tail -1 > $O/bench_of$of.json
  python - <<PY
import json
d=json.load(open("$O/bench_of$of.json"))
print("own_fill=$of", d["value"], d["ms_per_step"], d["step_ms"]["p50"], d["step_ms"]["p99"])
PY
done
rm -rf $O/trace_nt
HT_TIMERS=0 rocprofv3 --kernel-trace -d $O/trace_nt -o t --output-format csv -- python tools/host_trace.py > $O/trace_nt.log 2>&1
python tools/timeline.py $O/trace_nt > $O/timeline_nt.txt
grep "median" $O/trace_nt.log
sed -n 2,62p $O/timeline_nt.txt
