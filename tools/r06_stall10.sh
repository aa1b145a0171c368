#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
python tools/stall_repro_torch.py copy,idle,copy > $O/stall_repro_torch.jsonl 2> $O/stall_repro_torch.err
cat $O/stall_repro_torch.jsonl | cut -c1-600
tail -n 3 $O/stall_repro_torch.err
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/pytest_gpu_2.txt
cat $O/pytest_gpu_2.txt
