#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
python tools/stall_hunt3.py 3 base,sync_phase,no_timing,sync_phase,base > $O/stall_hunt9.jsonl 2> $O/stall_hunt9.err
cat $O/stall_hunt9.jsonl | cut -c1-500
tail -n 3 $O/stall_hunt9.err
