#!/bin/bash
# round 5 working loop on the GPU box: parity of the split pipeline, split against fused, per-kernel times, stage profile
# usage: tools/r05_ab.sh <tag> [quick]
set -u
cd $GRAFT_REPO_ROOT
TAG=${1:-x}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r05${TAG}_parity.log 2>&1
tail -2 gpurun_out/r05${TAG}_parity.log
X="--no-cpu-baseline --no-extras --no-end-to-end --streams 1024 --seconds 10 --steps 2 --warmup 1"
for i in 1 2; do
  python bench.py $X 2>/dev/null | grep '^{"metric"' > gpurun_out/r05${TAG}_split_$i.json
  LAMEHIP_FUSED=1 python bench.py $X 2>/dev/null | grep '^{"metric"' > gpurun_out/r05${TAG}_fused_$i.json
done
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/r05${TAG}_*_?.json')):
    try:
        r=json.load(open(f))
        print(f, r['value'], r['config']['per_stream_x_realtime'], r['per_rank'][0]['kernel_ms_avg'], r['pipeline'], r['checked_against_oracle']['result'])
    except Exception as e:
        print(f, 'ERR', e)
PY
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt_r05$TAG -- python $GRAFT_REPO_ROOT/bench.py $X > $GRAFT_REPO_ROOT/gpurun_out/kt_r05$TAG.log 2>&1 )
f=$(find gpurun_out/kt_r05$TAG -name '*kernel_stats.csv' | head -1)
grep '"lh_' $f | cut -c1-110 | tee gpurun_out/r05${TAG}_kernel_stats.txt
rm -rf gpurun_out/kt_r05$TAG
if [ "${2:-}" != quick ]; then
LAMEHIP_LIB=deprecated-lame-mirror_amd/lamehip/liblamehip_prof.so python tools/stage_profile.py 1024 4 > gpurun_out/r05${TAG}_stage_profile.txt 2>&1
grep -v " 0    0.0%" gpurun_out/r05${TAG}_stage_profile.txt | head -40
fi
