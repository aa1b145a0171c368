#!/bin/bash
# round 6: the stall hunt's variants on the library WITH the block cache (contexts hand their buffers on instead of hipFree / hipMalloc)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
python tools/stall_hunt4.py 3 no_empty,base,no_empty,base > $O/stall_hunt7.jsonl 2> $O/stall_hunt7.err
cat $O/stall_hunt7.jsonl | cut -c1-700
tail -n 3 $O/stall_hunt7.err
python bench.py --quick --steps 30 > $O/bench_quick7.json 2> $O/bench_quick7.err; tail -3 $O/bench_quick7.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06/bench_quick7.json").read().strip().splitlines()[-1])
print('ms/step %.3f'%d['ms_per_step'], 'value %.3e'%d['value'], d['step_ms'], d['results_collected_per_step'])
print('roofline frac', d['roofline']['frac'], d['roofline']['avg_ms'], 'iso', d['roofline']['isolated'])
PY
