#!/bin/bash
# parity sweeps on the round's FINAL kernels (filter pass in the tally's epilogue, single-entry rows out of k4_enum_bits' sigma loop)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
( timeout 1200 python tools/fuzz_enum.py 420 700 2>&1 | grep -v amdgpu | tail -4
  timeout 900 python tools/fuzz_parity.py 8000 8100 2>&1 | grep -v amdgpu | tail -3
  timeout 900 python tools/fuzz_chain.py 5000 5120 2>&1 | grep -v amdgpu | tail -3
  timeout 600 python tools/fuzz_island.py 30 36 2>&1 | grep -v amdgpu | tail -2
  timeout 600 python tools/determinism.py 2>&1 | grep -v amdgpu | tail -4 ) > $O/fuzz_sweeps2.txt 2>&1
cat $O/fuzz_sweeps2.txt
