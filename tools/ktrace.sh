#!/bin/bash
# timeline of the last N kernel dispatches of a command: tools/ktrace.sh N -- <command...>
N=$1; shift; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/ktrace && rocprofv3 --kernel-trace -d /tmp/ktrace -o p --output-format csv -- "$@" > /tmp/ktrace.log 2>&1
python - "$N" <<'PY'
import csv, sys
rows = list(csv.DictReader(open("/tmp/ktrace/p_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[1]):]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %9.1f  %8.1f us  q%-3s grid %8s  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r.get("Grid_Size", "?"), r["Kernel_Name"][:60]))
PY
