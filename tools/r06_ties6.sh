#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "chain_ties or tie_arithmetic or tie_only or chain_kernel or both_scopes or grid" 2>&1 | tail -3
( timeout 900 python tools/fuzz_chain_ties.py 0 200 2>&1 | grep -v amdgpu | tail -5
  timeout 900 python tools/fuzz_chain.py 5000 5040 2>&1 | grep -v amdgpu | tail -3 ) 2>&1
for wl in c3 c4; do
  rm -rf $O/st_$wl
  rocprofv3 --kernel-trace --stats -d $O/st_$wl -o p --output-format csv -- python bench.py --quick --workload $wl --steps 40 --warmup 10 > $O/st_$wl.json 2>/dev/null
  python - <<PY
import csv, json
d=json.loads(open("$O/st_$wl.json").read().strip().splitlines()[-1])
print("$wl", d["ms_per_step"], d["step_ms"]["p50"])
for r in csv.DictReader(open("$O/st_$wl/p_kernel_stats.csv")):
    if "k4_chain_wg" in r["Name"] or "k4_post<1024>" in r["Name"] or "k4_enum_bits" in r["Name"]: print("  ", r["Name"].replace("(anonymous namespace)::","")[:40], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
done
