#!/bin/bash
# bench line + rocprofv3 kernel stats of the same command (profiles/ artifacts)
TAG=${1:-r01_v6}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
tail -c 600 gpurun_out/$TAG/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG/prof -o p --output-format csv -- python bench.py --no-cpu-baseline > gpurun_out/$TAG/bench_prof.log 2>&1
cp gpurun_out/$TAG/prof/p_kernel_stats.csv gpurun_out/$TAG/kernel_stats.csv
cut -c1-110 gpurun_out/$TAG/kernel_stats.csv | head -30
cat gpurun_out/$TAG/bench.json | cut -c1-300
