#!/bin/bash
# per-kernel times of the phase stage with the restarts as bit states (LCR_ENUM_BITS=1) and without: tools/kstat_bits.sh [workload]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-c3}
for v in 0 1; do
rm -rf /tmp/ks; LCR_ENUM_BITS=$v rocprofv3 --kernel-trace --stats -d /tmp/ks -o p --output-format csv -- python bench.py --no-extras --no-cpu-baseline --no-traffic --sync-phase --workload $W --steps 30 > /tmp/ks.log 2>&1
echo "== enum_bits=$v"; python - <<'PY'
import csv,glob
for f in glob.glob("/tmp/ks/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f))):
        n=r["Name"].replace("(anonymous namespace)::","").replace("void ","")
        if n.startswith("k4_"): print("  %-50s calls %5s avg_us %9.1f" % (n[:50], r["Calls"], float(r["AverageNs"])/1e3))
PY
tail -1 /tmp/ks.log | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step %.3f'%d['ms_per_step'], {k:round(v,3) for k,v in d['stages']['api_ms'].items()})
except Exception as e: print(e)
"
done
