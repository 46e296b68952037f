#!/bin/bash
# first GPU trip of round 5: parity of the split pipeline, then split against fused on one box
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r05a_parity.log 2>&1
tail -5 gpurun_out/r05a_parity.log
X="--no-cpu-baseline --no-extras --no-end-to-end --streams 1024 --seconds 10 --steps 2 --warmup 1"
for i in 1 2; do
  python bench.py $X 2>/dev/null | grep '^{"metric"' > gpurun_out/r05a_split_$i.json
  LAMEHIP_FUSED=1 python bench.py $X 2>/dev/null | grep '^{"metric"' > gpurun_out/r05a_fused_$i.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05a_*_?.json')):
    try:
        r=json.load(open(f))
        print(f, r['value'], r['config']['per_stream_x_realtime'], r['pipeline'], r['checked_against_oracle'])
    except Exception as e:
        print(f, 'ERR', e)
PY
