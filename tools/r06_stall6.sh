#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
python tools/stall_hunt4.py 3 no_empty,ctx_first,create_sleep,warm_sleep,no_empty > $O/stall_hunt6.jsonl 2> $O/stall_hunt6.err
GPU_MAX_HW_QUEUES=4 python tools/stall_hunt4.py 3 no_empty >> $O/stall_hunt6.jsonl 2>> $O/stall_hunt6.err
cat $O/stall_hunt6.jsonl | cut -c1-700
tail -n 3 $O/stall_hunt6.err
