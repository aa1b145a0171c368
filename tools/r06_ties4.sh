#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_gpu_ties.txt
cat $O/pytest_gpu_ties.txt
( timeout 900 python tools/fuzz_parity.py 8100 8160 2>&1 | grep -v amdgpu | tail -3
  timeout 900 python tools/fuzz_chain.py 5080 5200 2>&1 | grep -v amdgpu | tail -3
  timeout 900 python tools/fuzz_chain_ties.py 400 900 2>&1 | grep -v amdgpu | tail -5
  timeout 600 python tools/fuzz_island.py 30 34 2>&1 | grep -v amdgpu | tail -2 ) > $O/fuzz_sweeps3.txt 2>&1
cat $O/fuzz_sweeps3.txt
