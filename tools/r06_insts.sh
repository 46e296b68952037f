#!/bin/bash
# GPU box: wave-instructions and wave-cycles per frame of the encode kernel for several libraries (one rocprofv3 --pmc pass each
# over bench 1024 x 10 s; 392 192 frames per launch)
cd $GRAFT_REPO_ROOT
X="--no-cpu-baseline --no-extras --no-end-to-end --streams 1024 --seconds 10 --steps 2 --warmup 1 --check-streams 4 --check-procs 1"
for L in "$@"; do
  ( cd /tmp && export TMPDIR=/tmp && LAMEHIP_LIB=$GRAFT_REPO_ROOT/deprecated-lame-mirror_amd/lamehip/$L timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_insts_$L -- python $GRAFT_REPO_ROOT/bench.py $X > $GRAFT_REPO_ROOT/gpurun_out/pmc_insts_$L.log 2>&1 )
  echo "== $L"; python tools/pmc_summary.py gpurun_out/pmc_insts_$L lh_encode | grep -v "^kernel"; grep '^{"metric"' gpurun_out/pmc_insts_$L.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'], d['pipeline'].get('kernels_ms_avg'))"
  rm -rf gpurun_out/pmc_insts_$L
done
