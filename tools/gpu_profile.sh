#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace statistics and PMC passes of the
# bench command; raw output under gpurun_out/, text summaries under gpurun_out/summ_*.txt
# (copy the ones to keep into profiles/).
# usage: tools/gpu_profile.sh <tag> [bench args...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
ARGS="${*:---streams 1024 --seconds 5 --steps 2 --warmup 1 --no-cpu-baseline --no-extras} --no-end-to-end --check-streams 4 --check-procs 1"   # (no child processes under the profiler: it attaches to each)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "${TRAFFIC_ONLY:-0}" = 1 ]; then
  # only the two memory-side passes: bytes the L2 exchanged with the fabric per launch
  for set in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$set -- python $ROOT/bench.py $ARGS > $OUT/pmc_${TAG}_$set.log 2>&1
    python $ROOT/tools/pmc_summary.py $OUT/pmc_${TAG}_$set lh_
    rm -rf $OUT/pmc_${TAG}_$set
  done > $OUT/summ_${TAG}_traffic.txt
  grep '^{"metric"' $OUT/pmc_${TAG}_WRITE_SIZE.log | tail -1 | cut -c1-300 >> $OUT/summ_${TAG}_traffic.txt
  cat $OUT/summ_${TAG}_traffic.txt
  exit 0
fi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$TAG -- python $ROOT/bench.py $ARGS > $OUT/kt_$TAG.log 2>&1
f=$(find $OUT/kt_$TAG -name '*kernel_stats.csv' | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py $ARGS"; [ -n "$f" ] && grep '^"Name\|^"lh_' "$f"; grep '^{"metric"' $OUT/kt_$TAG.log | tail -1; } > $OUT/summ_${TAG}_kernel_stats.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH" \
           "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$i -- python $ROOT/bench.py $ARGS > $OUT/pmc_${TAG}_$i.log 2>&1
done
{ echo "# rocprofv3 --pmc <set> --kernel-trace -- python bench.py $ARGS   (one pass per set; sums over the launches of the run)";
  for j in 1 2 3 4 5 6; do python $ROOT/tools/pmc_summary.py $OUT/pmc_${TAG}_$j lh_; done; } > $OUT/summ_${TAG}_pmc.txt
rm -rf $OUT/kt_$TAG $OUT/pmc_${TAG}_[0-9]   # raw traces are large; the summaries are what is kept
python $ROOT/tools/pmc_to_json.py $OUT/summ_${TAG}_pmc.txt $OUT/summ_${TAG}_kernel_stats.txt "bench.py $ARGS" > $OUT/summ_${TAG}_pmc.json
cat $OUT/summ_${TAG}_kernel_stats.txt | cut -c1-250 | head -12
cat $OUT/summ_${TAG}_pmc.txt
