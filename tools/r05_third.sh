#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_batch_every_stream" 2>&1 | tail -2; done
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r05c_parity.log 2>&1
tail -3 gpurun_out/r05c_parity.log
LAMEHIP_LIB=deprecated-lame-mirror_amd/lamehip/liblamehip_prof.so python tools/stage_profile.py 1024 4 > gpurun_out/r05c_stage_profile.txt 2>&1
head -42 gpurun_out/r05c_stage_profile.txt
