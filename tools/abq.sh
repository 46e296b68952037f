#!/bin/bash
# GPU box: A/B several builds on the same box, alternating, reporting the encode kernel's time too.  usage: tools/abq.sh <rounds> <lib1.so> ...
R=$1; shift
ARGS=${LAMEHIP_ABN_ARGS:---streams 1024 --seconds 10 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-end-to-end}
for i in $(seq $R); do
  for L in "$@"; do
    LAMEHIP_LIB=$PWD/deprecated-lame-mirror_amd/lamehip/$L python bench.py $ARGS 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['pipeline'].get('kernels_ms_avg'), d['checked_against_oracle']['result'])"
  done
done
