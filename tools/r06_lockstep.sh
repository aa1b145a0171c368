#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "enumeration or tie_only or tie_arithmetic or smoke or strand or composition or deep_region or edge" 2>&1 | tail -3
timeout 600 python tools/fuzz_enum.py 300 360 2>&1 | grep -v amdgpu | tail -3
for wl in c3 c4; do
  rm -rf $O/sl
  rocprofv3 --kernel-trace --stats -d $O/sl -o p --output-format csv -- python bench.py --quick --workload $wl --steps 40 --warmup 10 > $O/sl.json 2>/dev/null
  python - <<PY
import csv, json
d=json.loads(open("$O/sl.json").read().strip().splitlines()[-1])
out=["$wl step %.3f p50 %.3f" % (d["ms_per_step"], d["step_ms"]["p50"])]
for r in csv.DictReader(open("$O/sl/p_kernel_stats.csv")):
    if "k4_enum_bits" in r["Name"] or "k4_enum_reg" in r["Name"]: out.append("%s avg %.0f us" % (r["Name"].replace("(anonymous namespace)::","")[:14], float(r["AverageNs"])/1e3))
print(" | ".join(out))
PY
done
