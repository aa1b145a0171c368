#!/bin/bash
# PMC counters of the phase stage's kernels on the headline workload (synchronous steps): tools/pmc_phase.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for CTR in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/kpmc
  rocprofv3 --kernel-trace --pmc $CTR -d /tmp/kpmc -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --prewarm 2 --no-extras --no-cpu-baseline --no-traffic --sync-phase > /tmp/kpmc.log 2>&1
  python - <<'PY'
import csv, re, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("/tmp/kpmc/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if re.search("k4_", r["Kernel_Name"]):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:28]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, d in sorted(agg.items()):
    print("%-28s" % k, "launches", len(n[k]), " ".join("%s=%.4g" % (c, v / len(n[k])) for c, v in sorted(d.items())))
PY
done
