#!/bin/bash
# same-box A/B: working tree (k2_hist with the four-deep chain, k4_enum_bits with one any-tie test per row end) against HEAD
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "not eight_ranks and not c5_full and not two_ranks" 2>&1 | tail -4
for i in 1 2; do
  LCR_LIB=$PWD/gpurun_in/liblcr_head.so python bench.py --quick --steps 60 > $O/ab1_old_$i.json 2>/dev/null
  python bench.py --quick --steps 60 > $O/ab1_new_$i.json 2>/dev/null
done
python tools/ab_cmp.py $O/ab1_old_1.json $O/ab1_new_1.json $O/ab1_old_2.json $O/ab1_new_2.json
for i in 1; do
  LCR_LIB=$PWD/gpurun_in/liblcr_head.so python bench.py --quick --steps 30 --workload c4 > $O/ab1_c4_old_$i.json 2>/dev/null
  python bench.py --quick --steps 30 --workload c4 > $O/ab1_c4_new_$i.json 2>/dev/null
done
python tools/ab_cmp.py $O/ab1_c4_old_1.json $O/ab1_c4_new_1.json
timeout 900 python tests/golden/make_c5_golden.py 200 --mode f64 2>&1 | tail -2 | cut -c1-3000
