#!/bin/bash
# A/B on one box: fused filter pass (default) vs k2_filter; the record-free tiles early on a second queue (LCR_BG_TILES=-1); host laps of lcr_phase
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "filter_pass or random_cigar or demo or preset or c3_full or pileup" 2>&1 | tail -3
for rep in 1 2; do
for v in base nofuse early; do
  unset LCR_FUSE_FILTER LCR_BG_TILES
  [ $v = nofuse ] && export LCR_FUSE_FILTER=0
  [ $v = early ] && export LCR_BG_TILES=-1
  python bench.py --quick --steps 60 > $O/ab3_${v}_$rep.json 2>/dev/null
done; done
unset LCR_FUSE_FILTER LCR_BG_TILES
python tools/ab_cmp.py $O/ab3_base_1.json $O/ab3_nofuse_1.json $O/ab3_early_1.json $O/ab3_base_2.json $O/ab3_nofuse_2.json $O/ab3_early_2.json
for v in base early; do
  unset LCR_BG_TILES; [ $v = early ] && export LCR_BG_TILES=-1
  python bench.py --quick --steps 30 --workload c4 > $O/ab3_c4_${v}.json 2>/dev/null
done
unset LCR_BG_TILES
python tools/ab_cmp.py $O/ab3_c4_base.json $O/ab3_c4_early.json
LCR_PHASE_PROF=1 python bench.py --quick --steps 3 --warmup 1 --prewarm 3 --sync-phase 2>&1 >/dev/null | grep "^\[phase\]" | tail -12
