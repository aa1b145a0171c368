#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python tools/r06_seed377.py 377 2>&1 | grep -v amdgpu.ids | tail -6
timeout 900 python tools/fuzz_enum.py 300 420 2>&1 | grep -v amdgpu | tail -3
timeout 900 python -m pytest tests -m gpu -x -q -k "enum or tie or c3_full or c4 or deep_region or fallback or threshold or 65536" 2>&1 | tail -3
python bench.py --quick --steps 40 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step %.3f'%d['ms_per_step'], d['stages']['api_ms'])"
python bench.py --quick --steps 20 --workload c4 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 ms/step %.3f'%d['ms_per_step'], d['stages']['api_ms'])"
