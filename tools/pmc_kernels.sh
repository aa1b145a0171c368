cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
rm -rf gpurun_out/pmc2; mkdir -p gpurun_out/pmc2
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pmc2 -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc2/log.txt 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc2/*counter_collection.csv')
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen=set()
for row in csv.DictReader(open(f[0])):
    k = row['Kernel_Name'][:28]
    agg[k][row['Counter_Name']] += float(row['Counter_Value'])
    key=(k,row['Dispatch_Id']) 
    if key not in seen: seen.add(key); n[k]+=1
for k, v in agg.items():
    if any(s in k for s in ['k2_hist','k3_walk','k0_bin','k1_pileup','k4_enum_reg','k4_stage']):
        print(k, n[k], {a: round(b/n[k]) for a, b in v.items()})
PY
