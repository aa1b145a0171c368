cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o p --output-format csv -- python tools/c5_run.py --repeat 2 > /tmp/kt.log 2>&1
python - <<'PY'
import csv,glob
for f in glob.glob("/tmp/kt/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:8]:
        print("%-60s calls %5s avg_ms %9.3f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e6))
PY
tail -1 /tmp/kt.log | cut -c1-400
