#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
python tools/stall_hunt5.py 3 no_empty > $O/stall_hunt5.jsonl 2> $O/stall_hunt5.err
cat $O/stall_hunt5.jsonl | cut -c1-900
tail -n 3 $O/stall_hunt5.err
dmesg 2>/dev/null | tail -5
cat /proc/sys/kernel/numa_balancing /sys/kernel/mm/transparent_hugepage/enabled 2>/dev/null
