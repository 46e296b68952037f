for L in ${LIBS:-head tag head tag}; do
LAMEHIP_LIB=$PWD/deprecated-lame-mirror_amd/lamehip/liblamehip_$L.so python bench.py --streams 1024 --seconds 10 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); e=d['end_to_end']; print('$L', d['value'], 'e2e host', e['value'], 'dev', e['device_packed']['value'], 'resident', e['hbm_resident_same_sample'], e['device_packed']['bytes_checked_against_host_packer']['result'])"
done
