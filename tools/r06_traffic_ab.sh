#!/bin/bash
# GPU box: time (alternating) and HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of several builds.  usage: tools/r06_traffic_ab.sh <lib.so> ...
cd $GRAFT_REPO_ROOT
bash tools/abq.sh 2 "$@"
for L in "$@"; do
  echo "== traffic of $L"
  LAMEHIP_LIB=$PWD/deprecated-lame-mirror_amd/lamehip/$L TRAFFIC_ONLY=1 bash tools/gpu_profile.sh t_${L%.so} --streams 1024 --seconds 5 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep -v "^{" | cut -c1-200
done
