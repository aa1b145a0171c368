#!/bin/bash
# round 6 re-entry: the checkpointed build's GPU suite + the bench line (baseline for this session)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu_base.txt
cat $O/pytest_gpu_base.txt
python bench.py > $O/bench_base.json 2> $O/bench_base.err
tail -3 $O/bench_base.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06/bench_base.json").read().strip().splitlines()[-1])
print('ms/step %.3f'%d['ms_per_step'], 'value %.3e'%d['value'])
print(json.dumps(d['roofline'])[:1500])
print(json.dumps(d['stages'].get('kernel_ms'))[:1500])
print(json.dumps(d['stages'].get('api_ms'))[:600])
print(json.dumps(d['stages'].get('seeds'))[:1500])
PY
