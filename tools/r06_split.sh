#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "smoke or strand or edge or composition or two_contig or pipelined or chain_kernel or both_scopes or grid or fallback or deep_region or enumeration" 2>&1 | tail -3
for sc in 1 0 1 0; do
  for wl in c3 c4; do
    LCR_STAGE_SPLIT=$sc timeout 600 python bench.py --quick --workload $wl --steps 80 --warmup 10 2>/dev/null | tail -1 > $O/bs.json
    python - <<PY
import json
d=json.load(open("$O/bs.json"))
print("stage_split=$sc $wl step %.3f p50 %.3f p99 %.3f" % (d["ms_per_step"], d["step_ms"]["p50"], d["step_ms"]["p99"]))
PY
  done
done
