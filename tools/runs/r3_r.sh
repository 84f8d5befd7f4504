#!/bin/bash
export HIPIE_MIOPEN_FIND=0
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 ) 2>&1 | tail -9
