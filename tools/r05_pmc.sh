#!/bin/bash
# PMC passes over the split pipeline's kernels (bench at 1024 x 10 s); usage: tools/r05_pmc.sh <tag>
set -u
cd $GRAFT_REPO_ROOT
TAG=${1:-x}
X="--no-cpu-baseline --no-extras --no-end-to-end --streams 1024 --seconds 10 --steps 2 --warmup 1"
python bench.py $X 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; r=json.load(sys.stdin); print(r['value'], r['pipeline'], r['checked_against_oracle']['result'])"
( cd /tmp && export TMPDIR=/tmp && for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_r05$TAG -- python $GRAFT_REPO_ROOT/bench.py $X > /dev/null 2>&1
  for k in lh_attack_kernel lh_attack_scan lh_analysis lh_subband lh_encode; do python $GRAFT_REPO_ROOT/tools/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/pmc_r05$TAG $k | grep -v "^kernel"; done
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_r05$TAG
done ) > gpurun_out/r05${TAG}_pmc.txt 2>&1
grep -v "lh_encode\|attack" gpurun_out/r05${TAG}_pmc.txt | cut -c1-130
