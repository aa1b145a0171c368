#!/bin/bash
# Round-2 evidence for profiles/: bench lines, rocprofv3 kernel stats of the same commands, PMC passes (separate runs:
# SQ counters of K0 / K1 / K2 / K3 / K4 kernels; FETCH_SIZE and WRITE_SIZE of the pileup kernels).
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --no-cpu-baseline --no-c5 --profile ont-drna > $O/bench_ont-drna.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --no-c5 --workload c4 --steps 30 > $O/bench_c4_one_gpu.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python bench.py --no-cpu-baseline --no-c5 > $O/prof.log 2>&1
cp $O/prof/p_kernel_stats.csv $O/bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $O/prof_drna -o p --output-format csv -- python bench.py --no-cpu-baseline --no-c5 --profile ont-drna > $O/prof_drna.log 2>&1
cp $O/prof_drna/p_kernel_stats.csv $O/bench_ont-drna_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o p --output-format csv -- python tools/c5_run.py --repeat 2 > $O/c5_run.json 2> $O/prof_c5.log
cp $O/prof_c5/p_kernel_stats.csv $O/c5_kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_sq -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --prewarm 2 --no-cpu-baseline --no-c5 > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES -d $O/pmc_lds -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --prewarm 2 --no-cpu-baseline --no-c5 > $O/pmc_lds.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --prewarm 2 --no-cpu-baseline --no-c5 > $O/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --prewarm 2 --no-cpu-baseline --no-c5 > $O/pmc_w.log 2>&1
python - <<PY > $O/pmc_summary.txt
import csv, glob, collections
print("rocprofv3 --pmc passes over: python bench.py --steps 2 --warmup 1 --prewarm 2 --no-cpu-baseline --no-c5  (C3: 400 regions x 25 kb ONT-cDNA, 40x)")
print("per-launch averages; SQ_* = wave-level counts summed over the device; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them")
print("(MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts 64 B per 128-B request -> double it for wide coalesced reads; WRITE_SIZE as is)")
for tag in ("pmc_sq", "pmc_lds", "pmc_f", "pmc_w"):
    f = glob.glob("$O/%s/*counter_collection.csv" % tag)
    if not f:
        print(tag, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = set(); n = collections.Counter()
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"].split("(")[0][-40:]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (k, row["Dispatch_Id"])
        if key not in seen:
            seen.add(key); n[k] += 1
    print("==", tag)
    for k in sorted(agg):
        if any(s in k for s in ["k0_bin", "k1_pileup", "k2_hist", "k2_filter", "k3_walk", "k4_enum_reg", "k4_stage", "k4_chain", "k4_post", "scan_phase"]):
            print("  %-42s launches %3d  " % (k, n[k]) + "  ".join("%s=%.4g" % (a, b / n[k]) for a, b in sorted(agg[k].items())))
PY
cat $O/pmc_summary.txt | cut -c1-260
head -c 600 $O/bench.json; echo
cut -c1-120 $O/bench_kernel_stats.csv | head -14
