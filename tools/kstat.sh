#!/bin/bash
# per-kernel times of one command: tools/kstat.sh <pattern> -- <command...>
PAT=$1; shift; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/kstat && rocprofv3 --kernel-trace --stats -d /tmp/kstat -o p --output-format csv -- "$@" > /tmp/kstat.log 2>&1
python - "$PAT" <<'PY'
import csv, sys, re
for r in csv.DictReader(open("/tmp/kstat/p_kernel_stats.csv")):
    if re.search(sys.argv[1], r["Name"]): print("%-50s calls %5s avg %9.1f us  %5s %%" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
