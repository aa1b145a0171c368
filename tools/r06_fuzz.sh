#!/bin/bash
# parity sweeps on the round's kernels (k2_hist with the four-deep chain, the block cache, lcr_collect_phase) + the GPU suite three times
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
( timeout 900 python tools/fuzz_parity.py 7000 7120 2>&1 | tail -3
  timeout 900 python tools/fuzz_chain.py 4000 4160 2>&1 | tail -3
  timeout 600 python tools/fuzz_enum.py 300 420 2>&1 | tail -3
  timeout 600 python tools/fuzz_island.py 20 26 2>&1 | tail -3 ) > $O/fuzz_sweeps.txt 2>&1
cat $O/fuzz_sweeps.txt
for i in 1 2 3; do timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -2; done | tee $O/pytest_gpu_x3.txt
