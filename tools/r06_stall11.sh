#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
echo '--- HSA_ENABLE_SDMA=0' > $O/stall_hunt11.jsonl
HSA_ENABLE_SDMA=0 python tools/stall_hunt4.py 4 base >> $O/stall_hunt11.jsonl 2> $O/stall_hunt11.err
echo '--- HSA_ENABLE_INTERRUPT=0' >> $O/stall_hunt11.jsonl
HSA_ENABLE_INTERRUPT=0 python tools/stall_hunt4.py 4 base >> $O/stall_hunt11.jsonl 2>> $O/stall_hunt11.err
echo '--- default' >> $O/stall_hunt11.jsonl
python tools/stall_hunt4.py 4 base >> $O/stall_hunt11.jsonl 2>> $O/stall_hunt11.err
cat $O/stall_hunt11.jsonl | cut -c1-500
tail -n 3 $O/stall_hunt11.err
timeout 900 python -m pytest tests -m gpu -x -q -k "eight_ranks" 2>&1 | tail -5
timeout 600 python tests/golden/make_c5_golden.py 200 --mode f64 2>&1 | tail -3 | cut -c1-2500
