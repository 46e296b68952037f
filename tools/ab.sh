#!/bin/bash
# Run on the GPU box (through gpurun): A/B two builds of the library on the same box, alternating.
# usage: tools/ab.sh <libA.so> <libB.so> [rounds] [bench args...]
A=$1; B=$2; R=${3:-2}; shift 3 2>/dev/null
ARGS=${*:---streams 1024 --seconds 10 --steps 3 --warmup 1 --no-cpu-baseline --no-extras}
for i in $(seq $R); do
  for v in A B; do
    if [ $v = A ]; then L=$A; else L=$B; fi
    LAMEHIP_LIB=$PWD/$L python bench.py $ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
  done
done
