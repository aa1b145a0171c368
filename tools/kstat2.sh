#!/bin/bash
# per-kernel times of one command, top N: tools/kstat2.sh N -- <command...>
N=$1; shift; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/kstat && rocprofv3 --kernel-trace --stats -d /tmp/kstat -o p --output-format csv -- "$@" > /tmp/kstat.log 2>&1
python - "$N" <<'PY'
import csv, sys
rows = list(csv.DictReader(open("/tmp/kstat/p_kernel_stats.csv")))
for r in rows[:int(sys.argv[1])]: print("%-60s calls %6s avg %9.1f us  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
