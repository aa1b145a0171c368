#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for v in base occ3 occ4 base occ3; do
  lib=""; [ $v != base ] && lib=$GRAFT_REPO_ROOT/tools/experiments/variants/liblcr_$v.so
  for wl in c3 c4; do
    rm -rf $O/sp
    LCR_LIB=$lib rocprofv3 --kernel-trace --stats -d $O/sp -o p --output-format csv -- python bench.py --quick --workload $wl --steps 30 --warmup 10 > $O/sp.json 2>/dev/null
    python - <<PY
import csv, json
d=json.loads(open("$O/sp.json").read().strip().splitlines()[-1])
out=["$v $wl step %.3f" % d["ms_per_step"]]
for r in csv.DictReader(open("$O/sp/p_kernel_stats.csv")):
    if "k4_enum_bits" in r["Name"]: out.append("enum_bits avg %.0f us" % (float(r["AverageNs"])/1e3))
print(" | ".join(out))
PY
  done
done
