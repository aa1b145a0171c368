#!/bin/bash
# PMC counters of the kernels matching a pattern: tools/kpmc.sh <pattern> "<counters>" -- <command...>
PAT=$1; CTR=$2; shift; shift; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/kpmc && rocprofv3 --kernel-trace --pmc $CTR -d /tmp/kpmc -o p --output-format csv -- "$@" > /tmp/kpmc.log 2>&1
python - "$PAT" <<'PY'
import csv, sys, re, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("/tmp/kpmc/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if re.search(sys.argv[1], r["Kernel_Name"]):
            agg[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"][:40]].add(r["Dispatch_Id"])
for k, d in agg.items():
    print(k, "launches", len(n[k]), " ".join("%s=%.4g" % (c, v / len(n[k])) for c, v in sorted(d.items())))
PY
