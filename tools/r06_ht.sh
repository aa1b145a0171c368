#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
LCR_HOST_TRACE=1 HT_TIMERS=0 python tools/host_trace.py 2> $O/ht.err | tail -1
grep "^\[host\]" $O/ht.err | tail -40 | head -6
