#!/bin/bash
# GPU box: the tests added in round 6 (full-batch parity of configs [2] and [4], the fused batch path, kernels alternating mid-stream)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_batch or fused_kernel or changes_kernels" 2>&1 | tail -5
