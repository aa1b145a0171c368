#!/bin/bash
# K1: the first record batch requested in front of the LDS set-up (product) against HEAD; and a measurement build without the tile's geometry loads
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "filter_pass or random_cigar or demo or preset or pileup or k0_" 2>&1 | tail -2
for rep in 1 2 3; do
  LCR_LIB=$PWD/gpurun_in/liblcr_head.so python tools/k0_check.py --time-only --c4 2>/dev/null | grep "^c[34]"
  python tools/k0_check.py --time-only --c4 2>/dev/null | grep "^c[34]" | sed 's/^/hoist  /'
  LCR_LIB=$PWD/gpurun_in/liblcr_fakemeta.so python tools/k0_check.py --time-only --c4 2>/dev/null | grep "^c[34]" | sed 's/^/fake   /'
done 2>&1 | tee $O/ab6_k1.txt
