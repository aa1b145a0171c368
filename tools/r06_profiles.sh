#!/bin/bash
# Round-6 evidence for profiles/: the bench line, rocprofv3 kernel stats of the same command (headline workload, C4 share, C5),
# PMC passes (separate runs: SQ + LDS counters of the pileup, K2, K3 and K4 kernels; FETCH_SIZE / WRITE_SIZE of the pileup kernels
# are taken by bench.py itself: roofline.traffic).
TAG=${1:-r06p}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
Q="--no-extras --no-cpu-baseline --no-traffic"
rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python bench.py $Q > $O/prof.log 2>&1
cp $O/prof/p_kernel_stats.csv $O/bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $O/prof_c4 -o p --output-format csv -- python bench.py $Q --workload c4 --steps 30 > $O/bench_c4_one_gpu.json 2> $O/prof_c4.log
cp $O/prof_c4/p_kernel_stats.csv $O/bench_c4_one_gpu_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o p --output-format csv -- python tools/c5_run.py --repeat 2 > $O/c5_run.json 2> $O/prof_c5.log
cp $O/prof_c5/p_kernel_stats.csv $O/c5_kernel_stats.csv
LCR_PHASE_PROF=1 python tools/c5_run.py --repeat 1 2>&1 >/dev/null | grep '^\[phase\]' > $O/c5_phase_prof.txt
tools/pmc_c5.sh > $O/c5_pmc.txt 2>&1
P="--steps 2 --warmup 1 --prewarm 2 $Q"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o p --output-format csv -- python bench.py $P > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/pmc_lds -o p --output-format csv -- python bench.py $P > $O/pmc_lds.log 2>&1
python - "$O" "$P" <<'PY' > $O/pmc_summary.txt
import csv, glob, collections
import sys
O, P = sys.argv[1], sys.argv[2]
print("rocprofv3 --pmc passes over: python bench.py %s  (C3: 400 distinct genes x 25 kb ONT-cDNA, 40x)" % P)
print("per-launch averages; SQ_* = wave-level counts summed over the device (WAVE_CYCLES / WAIT_* / ACTIVE_* in quad-cycles); waves per SIMD = SQ_WAVE_CYCLES x 4 / (SQ_BUSY_CYCLES / 32 x 1024)")
for tag in ("pmc_sq", "pmc_lds"):
    f = glob.glob("%s/%s/*counter_collection.csv" % (O, tag))
    if not f:
        print(tag, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = set(); n = collections.Counter()
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (k, row["Dispatch_Id"])
        if key not in seen:
            seen.add(key); n[k] += 1
    print("==", tag)
    for k in sorted(agg):
        if any(s in k for s in ["k0_ops", "k0_desc", "k1_", "k2_hist", "k2_filter", "k3_walk", "k3_hits", "k4_enum_reg", "k4_enum_bits", "k4_enum_redo", "k4_enum_resolve", "k4_stage", "k4_chain", "k4_post"]):
            print("  %-42s launches %3d  " % (k, n[k]) + "  ".join("%s=%.4g" % (a, b / n[k]) for a, b in sorted(agg[k].items())))
PY
cut -c1-200 $O/pmc_summary.txt
head -c 400 $O/bench.json; echo
cut -c1-120 $O/bench_kernel_stats.csv | head -14
