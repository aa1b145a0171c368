#!/bin/bash
# k4_chain_wg: where the tie handling's time goes (measurement builds, -DCHAIN_ABL: 1 no class-8 handling, 2 no second pass in the kernel, 4 no per-region tie counts)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for v in base abl1 abl2 abl3 abl7; do
  lib=""; [ $v != base ] && lib=$GRAFT_REPO_ROOT/tools/experiments/variants/liblcr_$v.so
  for wl in c3 c4; do
    rm -rf $O/sa
    LCR_LIB=$lib rocprofv3 --kernel-trace --stats -d $O/sa -o p --output-format csv -- python bench.py --quick --workload $wl --steps 30 --warmup 10 > /dev/null 2>&1
    python - <<PY
import csv
for r in csv.DictReader(open("$O/sa/p_kernel_stats.csv")):
    if "k4_chain_wg" in r["Name"]: print("$v $wl chain_wg avg %.0f min %.0f max %.0f us" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
  done
done
