#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "chain_ties or chain or c3_full or c4 or tie or enumeration_threshold" 2>&1 | tail -12
