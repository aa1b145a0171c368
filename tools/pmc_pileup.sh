#!/bin/bash
# PMC counters of the pileup stage's kernels: tools/pmc_pileup.sh [LCR_LIB]   (two counter passes, per-launch averages)
LIB=$1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for CTR in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/kpmc
  LCR_LIB=$LIB rocprofv3 --kernel-trace --pmc $CTR -d /tmp/kpmc -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --prewarm 2 --quick > /tmp/kpmc.log 2>&1
  python - <<'PY'
import csv, sys, re, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("/tmp/kpmc/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if re.search("k0_|k1_", r["Kernel_Name"]):
            agg[r["Kernel_Name"][:24]][r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"][:24]].add(r["Dispatch_Id"])
for k, d in sorted(agg.items()):
    print("%-24s" % k, "launches", len(n[k]), " ".join("%s=%.4g" % (c, v / len(n[k])) for c, v in sorted(d.items())))
PY
done
