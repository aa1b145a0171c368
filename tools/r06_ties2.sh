#!/bin/bash
# chain regions with ties of classes 2 / 4 resolved at workgroup scope: sweeps + what the second launch costs on C3 / C4
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
( timeout 900 python tools/fuzz_chain_ties.py 0 300 2>&1 | grep -v amdgpu | tail -8
  timeout 900 python tools/fuzz_chain.py 5000 5080 2>&1 | grep -v amdgpu | tail -3 ) > $O/fuzz_chain_ties.txt 2>&1
cat $O/fuzz_chain_ties.txt
for ct in 1 0; do
  LCR_CHAIN_TIES=$ct timeout 600 python bench.py --steps 60 --warmup 10 2>/dev/null | tail -1 > $O/bench_ct$ct.json
  python - <<PY
import json
d=json.load(open("$O/bench_ct$ct.json"))
print("chain_ties=$ct", d["value"], d["ms_per_step"], d.get("step_ms"), json.dumps(d.get("stages")), json.dumps(d.get("c4")), json.dumps(d.get("phase_stage")))
PY
done
