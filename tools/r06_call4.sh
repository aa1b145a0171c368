#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
python tools/stall_hunt3.py 2 > $O/stall_hunt_variants.jsonl 2> $O/stall_hunt_variants.err
GPU_MAX_HW_QUEUES=4 python tools/stall_hunt3.py 2 base,no_timing >> $O/stall_hunt_variants.jsonl 2>> $O/stall_hunt_variants.err
cat $O/stall_hunt_variants.jsonl
tail -n 3 $O/stall_hunt_variants.err
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu_1.txt
cat $O/pytest_gpu_1.txt
python bench.py --no-extras --no-cpu-baseline --no-traffic --steps 40 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step %.3f'%d['ms_per_step'], 'value %.3e'%d['value'], 'roofline', d['roofline']['frac'], d['roofline']['avg_ms'], 'iso', d['roofline']['isolated'], d['stages']['api_ms'], d['stages']['kernel_ms'])
"
