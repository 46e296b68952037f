#!/bin/bash
# split against fused for the other rate controls (the extras' workloads)
cd $GRAFT_REPO_ROOT
X="--no-cpu-baseline --no-extras --no-end-to-end --streams 1024 --steps 2 --warmup 1"
run() { python bench.py $X "$@" 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['pipeline'].get('kernels_ms_avg'), d['checked_against_oracle']['result'])"; }
for args in "--seconds 5 --vbr 2" "--seconds 5 --vbr 2 --vbr-old" "--seconds 10 --samplerate 22050 --brate 64" "--seconds 5 --samplerate 48000 --brate 320 --mode 1 --bursts 40" "--seconds 5 --abr 160"; do
  for i in 1 2; do
    echo -n "split [$args] "; run $args
    echo -n "fused [$args] "; LAMEHIP_FUSED=1 run $args
  done
done
