#!/bin/bash
# kernel-level A/B (rocprofv3 --kernel-trace --stats): working tree against HEAD~ library, C3 and the C4 share
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for wl in c3 c4; do
for lib in old new; do
  if [ $lib = old ]; then export LCR_LIB=$PWD/gpurun_in/liblcr_head.so; else unset LCR_LIB; fi
  rm -rf /tmp/ks; rocprofv3 --kernel-trace --stats -d /tmp/ks -o p --output-format csv -- python bench.py --quick --steps 40 --workload $wl > /dev/null 2>&1
  echo "== $wl $lib"
  python - <<'PY'
import csv
rows = sorted(csv.DictReader(open("/tmp/ks/p_kernel_stats.csv")), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print("%-42s calls %5s avg %9.1f us" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:42], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done; done 2>&1 | tee $O/ab2_kernels.txt
