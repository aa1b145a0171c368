#!/bin/bash
# same-box comparison of bench.py under different environment settings: tools/ab_env.sh "A=1" "A=2 B=3" ...   (each run twice, interleaved)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for cfg in "$@"; do env $cfg python bench.py --no-extras --no-cpu-baseline --no-traffic --steps 60 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s'%'$cfg', 'ms/step %.3f'%d['ms_per_step'], 'value %.3e'%d['value'], 'phase %.3f'%d['stages']['api_ms']['lcr_phase'])
"; done; done
