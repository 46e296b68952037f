#!/bin/bash
# GPU box: old VBR loop (-V2 with sfb21, -V4 without) bench lines for several libraries, alternating; then its parity tests
cd $GRAFT_REPO_ROOT
X="--no-cpu-baseline --no-extras --no-end-to-end --streams 1024 --seconds 5 --steps 2 --warmup 1"
for i in 1 2; do for L in "$@"; do
  for A in "--vbr 2 --vbr-old" "--vbr 4 --vbr-old"; do
    LAMEHIP_LIB=$PWD/deprecated-lame-mirror_amd/lamehip/$L python bench.py $X $A 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', '$A', d['value'], d['pipeline'].get('kernels_ms_avg'), d['checked_against_oracle']['result'])"
  done
done; done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "vbrold or vbr_old or old" 2>&1 | tail -2
