#!/bin/bash
# GPU box, round 6 baseline: bench (1024 x 10 s, two rounds), stage profile of the encode kernel, phase profile of the analysis kernels
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=deprecated-lame-mirror_amd/lamehip
bash tools/abq.sh 2 liblamehip.so > gpurun_out/r06a_abq.log 2>&1
LAMEHIP_LIB=$PWD/$L/liblamehip_prof.so timeout 300 python tools/stage_profile.py 1024 4 > gpurun_out/r06a_stage_profile.txt 2>&1
LAMEHIP_LIB=$PWD/$L/liblamehip_aprof.so timeout 300 python tools/an_profile.py 1024 4 > gpurun_out/r06a_an_profile.txt 2>&1
cat gpurun_out/r06a_abq.log; tail -30 gpurun_out/r06a_an_profile.txt
