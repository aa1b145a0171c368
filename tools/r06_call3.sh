#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
LD_PRELOAD=$PWD/gpurun_in/ioslow.so IOSLOW_MS=3 python tools/stall_hunt2.py 4 > $O/stall_hunt3.jsonl 2> $O/stall_hunt3.err
grep -c ioslow $O/stall_hunt3.err
grep '"slow": \[{' $O/stall_hunt3.jsonl | cut -c1-400
tail -1 $O/stall_hunt3.jsonl
grep -A 14 '^\[ioslow\]' $O/stall_hunt3.err | cut -c1-200 | head -150
