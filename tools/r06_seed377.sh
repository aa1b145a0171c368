#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python tools/r06_seed377.py 377 2>&1 | grep -v amdgpu.ids
LCR_ENUM_BITS=0 python tools/r06_seed377.py 377 2>&1 | grep -v amdgpu.ids
LCR_LIB=$PWD/gpurun_in/liblcr_head.so python tools/r06_seed377.py 377 2>&1 | grep -v amdgpu.ids
LCR_FUSE_FILTER=0 python tools/r06_seed377.py 377 2>&1 | grep -v amdgpu.ids
bash tools/r06_ab3.sh
