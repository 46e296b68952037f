#!/bin/bash
# GPU box: the end-to-end pipeline (device bit packer) of two builds in turn (base.so = the build before)
set -u
cd $GRAFT_REPO_ROOT
for L in ${LIBS:-base.so liblamehip.so base.so liblamehip.so}; do
  LAMEHIP_LIB=$PWD/deprecated-lame-mirror_amd/lamehip/$L python bench.py --streams 1024 --seconds 10 --steps 1 --warmup 1 --no-cpu-baseline --check-streams 4 --check-procs 2 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); e=d['end_to_end']; print('$L', d['value'], 'e2e host', e['value'], 'device', e['device_packed']['value'], 'steady', e['device_packed']['steady_state'], 'resident', e['hbm_resident_same_sample'], e['device_packed']['bytes_checked_against_host_packer']['result'])"
done
