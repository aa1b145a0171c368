#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "chain_ties or tie_arithmetic or tie_only or chain_kernel or both_scopes" 2>&1 | tail -3
( timeout 900 python tools/fuzz_chain_ties.py 0 300 2>&1 | grep -v amdgpu | tail -5
  timeout 900 python tools/fuzz_chain.py 5000 5040 2>&1 | grep -v amdgpu | tail -3 ) 2>&1
for ct in 1 0 1 0; do
  LCR_CHAIN_TIES=$ct timeout 600 python bench.py --quick --steps 100 --warmup 10 2>/dev/null | tail -1 > $O/bench_ct$ct.json
  python - <<PY
import json
d=json.load(open("$O/bench_ct$ct.json"))
print("chain_ties=$ct", d["value"], d["ms_per_step"], d["step_ms"])
PY
done
rm -rf $O/trace_nt
HT_TIMERS=0 rocprofv3 --kernel-trace -d $O/trace_nt -o t --output-format csv -- python tools/host_trace.py > $O/trace_nt.log 2>&1
python tools/timeline.py $O/trace_nt > $O/timeline_nt.txt
grep "median" $O/trace_nt.log
grep -c . $O/timeline_nt.txt
