#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
python tools/stall_hunt4.py 2 > $O/stall_hunt4.jsonl 2> $O/stall_hunt4.err
cat $O/stall_hunt4.jsonl
tail -n 3 $O/stall_hunt4.err
