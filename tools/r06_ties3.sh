#!/bin/bash
# class 8 at workgroup scope too: tests, sweeps, cost on C3 / C4
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "chain_ties or tie_arithmetic or tie_only or chain_kernel or both_scopes" 2>&1 | tail -5
( timeout 900 python tools/fuzz_chain_ties.py 0 400 2>&1 | grep -v amdgpu | tail -8
  timeout 900 python tools/fuzz_chain.py 5000 5080 2>&1 | grep -v amdgpu | tail -3 ) > $O/fuzz_chain_ties.txt 2>&1
cat $O/fuzz_chain_ties.txt
for ct in 1 0 1 0; do
  LCR_CHAIN_TIES=$ct timeout 600 python bench.py --steps 60 --warmup 10 2>/dev/null | tail -1 > $O/bench_ct$ct.json
  python - <<PY
import json
d=json.load(open("$O/bench_ct$ct.json"))
print("chain_ties=$ct", d["value"], d["ms_per_step"], d["stages"]["api_ms"], d["stages"]["c4_share"]["ms_per_step"], d["stages"]["c4_share"]["api_ms"])
PY
done
