#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for rl in 65536 0; do
rm -rf $O/trace_c4
LCR_REDO_LDS=$rl HT_WORKLOAD=c4 HT_TIMERS=0 rocprofv3 --kernel-trace -d $O/trace_c4 -o t --output-format csv -- python tools/host_trace.py > $O/trace_c4.log 2>&1
python tools/timeline.py $O/trace_c4 6 > $O/timeline_c4_$rl.txt
echo "redo_lds=$rl"; grep median $O/trace_c4.log
grep "k4_enum\|k4_post\|k1_zonefix\|k2_filter\|k0_ops\|k4_chain" $O/timeline_c4_$rl.txt | head -24
done
