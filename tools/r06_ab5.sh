#!/bin/bash
# A/B: non-temporal stores of the record-free tiles (default build) against plain stores (gpurun_in/liblcr_nont.so, -DK1_NT_STORES=0)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for rep in 1 2 3; do
  LCR_LIB=$PWD/gpurun_in/liblcr_nont.so python bench.py --quick --steps 60 > $O/ab5_plain_$rep.json 2>/dev/null
  python bench.py --quick --steps 60 > $O/ab5_nt_$rep.json 2>/dev/null
done
python tools/ab_cmp.py $O/ab5_plain_1.json $O/ab5_nt_1.json $O/ab5_plain_2.json $O/ab5_nt_2.json $O/ab5_plain_3.json $O/ab5_nt_3.json
LCR_LIB=$PWD/gpurun_in/liblcr_nont.so python bench.py --quick --steps 30 --workload c4 > $O/ab5_c4_plain.json 2>/dev/null
python bench.py --quick --steps 30 --workload c4 > $O/ab5_c4_nt.json 2>/dev/null
python tools/ab_cmp.py $O/ab5_c4_plain.json $O/ab5_c4_nt.json
