#!/bin/bash
# GPU box: A/B several builds by kernel time only (no checks: for experiments whose output is deliberately wrong), alternating.
# usage: tools/abt.sh <rounds> <lib1.so> <lib2.so> ...   (libraries relative to deprecated-lame-mirror_amd/lamehip/)
R=$1; shift
cd $GRAFT_REPO_ROOT
for i in $(seq $R); do
  for L in "$@"; do
    echo -n "$L: "; LAMEHIP_LIB=$PWD/deprecated-lame-mirror_amd/lamehip/$L python tools/time_kernel.py 1024 ${ABT_SECONDS:-6} 3 2>&1 | tail -1
  done
done
