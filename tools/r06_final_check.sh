#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu_final.txt
cat $O/pytest_gpu_final.txt
( timeout 900 python tools/fuzz_enum.py 420 620 2>&1 | grep -v amdgpu | tail -3
  timeout 900 python tools/fuzz_parity.py 8200 8260 2>&1 | grep -v amdgpu | tail -3
  timeout 600 python tools/fuzz_chain.py 5200 5280 2>&1 | grep -v amdgpu | tail -3
  timeout 600 python tools/fuzz_chain_ties.py 900 1100 2>&1 | grep -v amdgpu | tail -3
  timeout 600 python tools/fuzz_island.py 30 33 2>&1 | grep -v amdgpu | tail -2
  timeout 600 python tools/determinism.py 2>&1 | grep -v amdgpu | tail -4 ) > $O/fuzz_sweeps_final.txt 2>&1
cat $O/fuzz_sweeps_final.txt
