#!/bin/bash
# Profiling recipe used for profiles/ (run on the GPU box from the repo root):
#   kernel stats:  rocprofv3 --kernel-trace --stats -d OUT -o p --output-format csv -- python tools/k1time.py ont-cdna 0
#   HBM traffic :  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d OUT/f ... ; rocprofv3 --kernel-trace --pmc WRITE_SIZE -d OUT/w ...
# (PMC passes are separate runs, never combined with other trace domains.)
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(dirname "$(dirname "$(readlink -f "$0")")")}; cd "$R"
OUT=${1:-gpurun_out/prof}; PROFILE=${2:-ont-cdna}
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o p --output-format csv -- python tools/k1time.py "$PROFILE" 0 > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/f" -o p --output-format csv -- python tools/k1time.py "$PROFILE" 0 > "$OUT/f.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/w" -o p --output-format csv -- python tools/k1time.py "$PROFILE" 0 > "$OUT/w.log" 2>&1
head -8 "$OUT"/stats/p_kernel_stats.csv
