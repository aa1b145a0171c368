#!/bin/bash
# the enumeration branch's repair pass with its matrix in LDS (lcr_debug_set redo_lds): parity + the C4 share / C3 steps
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "tie_only or tie_arithmetic or enumeration_kernels or deep_region or fallback" 2>&1 | tail -3
for rl in 65536 0 65536 0; do
  for wl in c4 c3; do
    rm -rf $O/sr
    LCR_REDO_LDS=$rl rocprofv3 --kernel-trace --stats -d $O/sr -o p --output-format csv -- python bench.py --quick --workload $wl --steps 40 --warmup 10 > $O/sr.json 2>/dev/null
    python - <<PY
import csv, json
d=json.loads(open("$O/sr.json").read().strip().splitlines()[-1])
out=["redo_lds=$rl $wl step %.3f p50 %.3f" % (d["ms_per_step"], d["step_ms"]["p50"])]
for r in csv.DictReader(open("$O/sr/p_kernel_stats.csv")):
    if "k4_enum_redo" in r["Name"]: out.append("redo calls %s avg %.0f max %.0f us" % (r["Calls"], float(r["AverageNs"])/1e3, float(r["MaxNs"])/1e3))
print(" | ".join(out))
PY
  done
done
