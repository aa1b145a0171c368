#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for ct in 1 0; do
  rm -rf $O/st_$ct
  LCR_CHAIN_TIES=$ct rocprofv3 --kernel-trace --stats -d $O/st_$ct -o p --output-format csv -- python bench.py --quick --steps 60 --warmup 10 > /dev/null 2>&1
  echo "chain_ties=$ct"; grep "k4_chain_wg\|k4_post<1024>\|k4_enum_bits" $O/st_$ct/p_kernel_stats.csv | cut -c1-60,100-260 | sed 's/^/  /'
  python - <<PY
import csv
for r in csv.DictReader(open("$O/st_$ct/p_kernel_stats.csv")):
    if "k4_chain_wg" in r["Name"] or "k4_post<1024>" in r["Name"]: print("  ", r["Name"][:50], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
done
