#!/bin/bash
# steady-state steps of the headline workload as a kernel timeline (rocprofv3 --kernel-trace), with and without the stage timers' event records
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O; rm -rf $O/trace $O/trace_nt
HT_TIMERS=1 rocprofv3 --kernel-trace -d $O/trace -o t --output-format csv -- python tools/host_trace.py > $O/trace.log 2>&1
HT_TIMERS=0 rocprofv3 --kernel-trace -d $O/trace_nt -o t --output-format csv -- python tools/host_trace.py > $O/trace_nt.log 2>&1
python tools/timeline.py $O/trace > $O/timeline.txt
python tools/timeline.py $O/trace_nt > $O/timeline_nt.txt
tail -2 $O/trace.log $O/trace_nt.log
head -70 $O/timeline_nt.txt
