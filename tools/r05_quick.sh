#!/bin/bash
# GPU box: the split pipeline's parity on the full 1024-stream launch, then two bench lines (10 s streams); no profiler
set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_batch or golden" 2>&1 | tail -2
bash tools/abq.sh 2 liblamehip.so
