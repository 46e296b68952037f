#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r05b_parity.log 2>&1
tail -3 gpurun_out/r05b_parity.log
X="--no-cpu-baseline --no-extras --no-end-to-end --streams 1024 --seconds 10 --steps 2 --warmup 1"
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt_r05b -- python $GRAFT_REPO_ROOT/bench.py $X > $GRAFT_REPO_ROOT/gpurun_out/kt_r05b.log 2>&1 )
f=$(find gpurun_out/kt_r05b -name '*kernel_stats.csv' | head -1)
cat $f | cut -c1-200 | head -12 | tee gpurun_out/r05b_kernel_stats.txt
rm -rf gpurun_out/kt_r05b
LAMEHIP_LIB=deprecated-lame-mirror_amd/lamehip/liblamehip_prof.so python tools/stage_profile.py 1024 4 > gpurun_out/r05b_stage_profile.txt 2>&1
head -42 gpurun_out/r05b_stage_profile.txt
( cd /tmp && export TMPDIR=/tmp && for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH"; do
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_r05b -- python $GRAFT_REPO_ROOT/bench.py $X > /dev/null 2>&1
  for k in lh_attack_kernel lh_attack_scan lh_analysis lh_subband lh_encode; do echo "== $k"; python $GRAFT_REPO_ROOT/tools/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/pmc_r05b $k; done
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_r05b
done ) > gpurun_out/r05b_pmc.txt 2>&1
cat gpurun_out/r05b_pmc.txt | cut -c1-200
