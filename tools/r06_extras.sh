#!/bin/bash
# GPU box: the bench's extras-like lines for one or more libraries (CBR 320 at 48 kHz with bursts, CBR 192, ABR 160)
cd $GRAFT_REPO_ROOT
X="--no-cpu-baseline --no-extras --no-end-to-end --streams 1024 --seconds 5 --steps 2 --warmup 1"
for L in "$@"; do
  for A in "--samplerate 48000 --brate 320 --mode 1 --bursts 40" "--brate 192" "--abr 160" "--brate 128 --seconds 10"; do
    LAMEHIP_LIB=$PWD/deprecated-lame-mirror_amd/lamehip/$L python bench.py $X $A 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', '$A', d['value'], d['pipeline'].get('kernels_ms_avg'), d['checked_against_oracle']['result'])"
  done
done
