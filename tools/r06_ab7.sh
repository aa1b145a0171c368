#!/bin/bash
# k4_enum_bits with the rows of one entry in a pass of their own: parity (enumeration tests, fuzz_enum) and kernel times against HEAD
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "enum or tie or c3_full or c4 or deep_region or fallback or threshold" 2>&1 | tail -3
timeout 600 python tools/fuzz_enum.py 0 60 2>&1 | tail -2
for wl in c3 c4; do
for lib in old new; do
  if [ $lib = old ]; then export LCR_LIB=$PWD/gpurun_in/liblcr_head.so; else unset LCR_LIB; fi
  rm -rf /tmp/ks; rocprofv3 --kernel-trace --stats -d /tmp/ks -o p --output-format csv -- python bench.py --quick --steps 40 --workload $wl > $O/ab7_${wl}_$lib.json 2>/dev/null
  echo "== $wl $lib"
  python - <<'PY'
import csv
rows = sorted(csv.DictReader(open("/tmp/ks/p_kernel_stats.csv")), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:6]:
    print("%-42s calls %5s avg %9.1f us" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:42], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done; done 2>&1 | tee $O/ab7_kernels.txt
unset LCR_LIB
python tools/ab_cmp.py $O/ab7_c3_old.json $O/ab7_c3_new.json $O/ab7_c4_old.json $O/ab7_c4_new.json
