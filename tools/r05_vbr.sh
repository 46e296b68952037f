#!/bin/bash
# GPU box: primitives self-test + VBR parity, then VBR -V2 / -V5 bench lines
set -u
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "selftest or vbr or golden" -n 4 2>&1 | tail -2
for V in 2 5; do
LAMEHIP_ABN_ARGS="--streams 1024 --seconds 10 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-end-to-end --vbr $V" bash tools/abq.sh 2 ${LIBS:-liblamehip.so}
done
