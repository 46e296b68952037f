#!/bin/bash
# GPU box: what working through a launch in frame windows costs (lamehip.h: lamehip_batch_last_windows), at the driver's size.
# usage: tools/r06_windows.sh [windows' sizes in frames ...]   (0 = the whole launch at once)
cd ${GRAFT_REPO_ROOT:-.}
for W in ${@:-0 1149 575 288 144 0}; do
  LAMEHIP_MID_WINDOW=$W python bench.py --streams 1024 --seconds 60 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-end-to-end 2>/dev/null | grep '^{"metric"' \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('window', $W, 'k x', round(d['value']/1e3,2), 'ms/step', d['ms_per_step'], d['pipeline'].get('kernels_ms_avg'), d['checked_against_oracle']['result'])"
done
