#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
python tools/stall_hunt4.py 5 keep_ctx,same_dv,idle150,copy_idle,base > $O/stall_hunt8.jsonl 2> $O/stall_hunt8.err
cat $O/stall_hunt8.jsonl | cut -c1-700
tail -n 3 $O/stall_hunt8.err
