#!/bin/bash
# longer parity sweeps on the round's final kernels
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
( timeout 1500 python tools/fuzz_enum.py 1500 1900 2>&1 | grep -v amdgpu | tail -3
  timeout 1200 python tools/fuzz_parity.py 9200 9280 2>&1 | grep -v amdgpu | tail -3
  timeout 900 python tools/fuzz_chain.py 6200 6300 2>&1 | grep -v amdgpu | tail -3
  timeout 900 python tools/fuzz_chain_ties.py 2600 2900 2>&1 | grep -v amdgpu | tail -3 ) > $O/fuzz_soak.txt 2>&1
cat $O/fuzz_soak.txt
