#!/bin/bash
# GPU box: A/B several builds of the library on the same box, alternating.  usage: tools/abn.sh <rounds> <lib1.so> <lib2.so> ...
# (libraries relative to deprecated-lame-mirror_amd/lamehip/; LAMEHIP_ABN_ARGS overrides the bench arguments)
R=$1; shift
ARGS=${LAMEHIP_ABN_ARGS:---streams 1024 --seconds 10 --steps 3 --warmup 1 --no-cpu-baseline --no-extras}
for i in $(seq $R); do
  for L in "$@"; do
    LAMEHIP_LIB=$PWD/deprecated-lame-mirror_amd/lamehip/$L python bench.py $ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step'], d.get('checked_against_oracle'))"
  done
done
