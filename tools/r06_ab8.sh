#!/bin/bash
# HiFi presets: the poly-A pass and the record-free tiles in one launch (default) against two kernels (LCR_ZF_FUSED=0)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "poly or preset or c4 or demo or random_cigar or zone or two_ranks or driver" 2>&1 | tail -3
for rep in 1 2 3; do
  LCR_ZF_FUSED=0 python bench.py --quick --steps 30 --workload c4 > $O/ab8_c4_serial_$rep.json 2>/dev/null
  python bench.py --quick --steps 30 --workload c4 > $O/ab8_c4_overlap_$rep.json 2>/dev/null
done
python tools/ab_cmp.py $O/ab8_c4_serial_1.json $O/ab8_c4_overlap_1.json $O/ab8_c4_serial_2.json $O/ab8_c4_overlap_2.json $O/ab8_c4_serial_3.json $O/ab8_c4_overlap_3.json
