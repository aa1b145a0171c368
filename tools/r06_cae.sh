#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for sc in 1 0 1 0; do
  for wl in c3 c4; do
    LCR_CHAIN_AFTER_ENUM=$sc timeout 600 python bench.py --quick --workload $wl --steps 80 --warmup 10 2>/dev/null | tail -1 > $O/bs.json
    python - <<PY
import json
d=json.load(open("$O/bs.json"))
print("chain_after_enum=$sc $wl step %.3f p50 %.3f p99 %.3f" % (d["ms_per_step"], d["step_ms"]["p50"], d["step_ms"]["p99"]))
PY
  done
done
