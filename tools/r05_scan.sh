#!/bin/bash
# GPU box: per-kernel times of two builds (base.so = the build before)
set -u
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for L in ${LIBS:-base.so liblamehip.so}; do
  rm -rf /tmp/kt; mkdir -p /tmp/kt
  LAMEHIP_LIB=$ROOT/deprecated-lame-mirror_amd/lamehip/$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $ROOT/bench.py --streams 1024 --seconds ${SECS:-10} --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-end-to-end --check-streams 4 --check-procs 1 >/tmp/kt.log 2>&1
  echo "== $L"; f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && grep -E '^"lh_|^"Name' "$f" | cut -d, -f1-5 | head -8 || tail -5 /tmp/kt.log
done
