#!/bin/bash
# Run on the GPU box (through gpurun): the L2's own view of the encode kernel's write traffic --
# requests it received, lines it wrote back or evicted, requests it sent to the fabric.
# usage: tools/gpu_tcc.sh <tag> [bench args...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
ARGS=${*:---streams 1024 --seconds 5 --steps 2 --warmup 1 --no-cpu-baseline --no-extras}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "TCC_WRITE_sum TCC_WRITEBACK_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_REQ_sum TCC_READ_sum TCC_ATOMIC_sum TCC_STREAMING_REQ_sum" \
           "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_NC_WRITE_REQ_sum TCP_TCC_UC_WRITE_REQ_sum" \
           "TCP_TCC_CC_WRITE_REQ_sum TCP_TCC_RW_WRITE_REQ_sum SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/tcc_${TAG}_$n -- python $ROOT/bench.py $ARGS > $OUT/tcc_${TAG}_$n.log 2>&1
  python $ROOT/tools/pmc_summary.py $OUT/tcc_${TAG}_$n lh_encode || tail -3 $OUT/tcc_${TAG}_$n.log
  rm -rf $OUT/tcc_${TAG}_$n
done > $OUT/summ_${TAG}_tcc.txt 2>&1
cat $OUT/summ_${TAG}_tcc.txt
