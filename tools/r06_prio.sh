#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for cfg in "0 0" "1 1" "1 0" "0 1" "0 0" "1 1"; do
  set -- $cfg
  for wl in c3 c4; do
    LCR_PHASE_PRIO=$1 LCR_NO_GATE=$2 timeout 600 python bench.py --quick --workload $wl --steps 60 --warmup 10 2>/dev/null | tail -1 > $O/bp.json
    python - <<PY
import json
d=json.load(open("$O/bp.json"))
print("prio=$1 no_gate=$2 $wl step %.3f p50 %.3f p99 %.3f pileup_ms %.3f" % (d["ms_per_step"], d["step_ms"]["p50"], d["step_ms"]["p99"], d["roofline"]["avg_ms"]))
PY
  done
done
