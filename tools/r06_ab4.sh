#!/bin/bash
# A/B: the record-free tiles' stores in front of the tally (LCR_BG_TILES=-2) against behind it (default)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for rep in 1 2 3; do
for v in base first; do
  unset LCR_BG_TILES; [ $v = first ] && export LCR_BG_TILES=-2
  python bench.py --quick --steps 60 > $O/ab4_${v}_$rep.json 2>/dev/null
done; done
unset LCR_BG_TILES
python tools/ab_cmp.py $O/ab4_base_1.json $O/ab4_first_1.json $O/ab4_base_2.json $O/ab4_first_2.json $O/ab4_base_3.json $O/ab4_first_3.json
for v in base first; do
  unset LCR_BG_TILES; [ $v = first ] && export LCR_BG_TILES=-2
  python bench.py --quick --steps 30 --workload c4 > $O/ab4_c4_${v}.json 2>/dev/null
done
unset LCR_BG_TILES
python tools/ab_cmp.py $O/ab4_c4_base.json $O/ab4_c4_first.json
LCR_PHASE_PROF=1 python bench.py --quick --steps 3 --warmup 1 --prewarm 3 --sync-phase 2>&1 >/dev/null | grep "^\[phase\] [a-z]" | tail -8
