#!/bin/bash
# Run on the GPU box (through gpurun): arbitrary PMC sets over the bench command, one pass per set.
# usage: tools/gpu_pmc_sets.sh <tag> "<set 1>" "<set 2>" ...   (bench args from $BENCH_ARGS; kernels whose name contains $KFILTER, default lh_encode)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
ARGS="${BENCH_ARGS:---streams 1024 --seconds 5 --steps 2 --warmup 1 --no-cpu-baseline --no-extras} --no-end-to-end --check-streams 4 --check-procs 1"   # (no child processes under the profiler: it attaches to each and the run does not end)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/ps_${TAG}_$i -- python $ROOT/bench.py $ARGS > $OUT/ps_${TAG}_$i.log 2>&1
  python $ROOT/tools/pmc_summary.py $OUT/ps_${TAG}_$i ${KFILTER:-lh_encode} || tail -3 $OUT/ps_${TAG}_$i.log
  rm -rf $OUT/ps_${TAG}_$i
done > $OUT/summ_${TAG}_sets.txt 2>&1
cat $OUT/summ_${TAG}_sets.txt
