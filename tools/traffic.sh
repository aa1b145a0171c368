cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
rm -rf gpurun_out/traffic; mkdir -p gpurun_out/traffic
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/traffic/f -o p --output-format csv -- python tools/k1time.py ont-cdna 0 > gpurun_out/traffic/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/traffic/w -o p --output-format csv -- python tools/k1time.py ont-cdna 0 > gpurun_out/traffic/w.log 2>&1
python - <<'PY'
import csv, glob, collections
for tag in ("f", "w"):
    f = glob.glob('gpurun_out/traffic/%s/*counter_collection.csv' % tag)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f[0])):
        k = (row['Kernel_Name'][:30], row['Counter_Name'])
        agg[k][0] += float(row['Counter_Value']); agg[k][1] += 1
    for k, v in sorted(agg.items()):
        if 'k1_pileup' in k[0] or 'k0_bin' in k[0] or 'hpmask' in k[0]:
            print(tag, k, "sum", v[0], "launches", v[1], "per launch", v[0] / v[1])
PY
