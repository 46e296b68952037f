#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/r03_collect.sh'): every profile and bench line kept under profiles/r03_* from ONE
# build -- kernel-trace statistics and PMC passes of the default, VBR, old-VBR and four-wave runs (summaries under
# gpurun_out/summ_*), stage profiles of the LH_PROF build (make -C deprecated-lame-mirror_amd/csrc prof first), then the
# bench lines with the fresh PMC records in place so that they quote them.
set -u
cd $GRAFT_REPO_ROOT
S="--streams 1024 --seconds 5 --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
bash tools/gpu_profile.sh r03 > gpurun_out/log_r03.txt 2>&1
bash tools/gpu_profile.sh r03_vbr2 $S --vbr 2 > gpurun_out/log_r03_vbr2.txt 2>&1
bash tools/gpu_profile.sh r03_vbrold2 $S --vbr 2 --vbr-old > gpurun_out/log_r03_vbrold2.txt 2>&1
LAMEHIP_KERNEL_WAVES=4 bash tools/gpu_profile.sh r03_waves4 > gpurun_out/log_r03_waves4.txt 2>&1
cp gpurun_out/summ_r03_pmc.json profiles/r03_pmc.json
cp gpurun_out/summ_r03_vbr2_pmc.json profiles/r03_pmc_vbr2.json
cp gpurun_out/summ_r03_vbrold2_pmc.json profiles/r03_pmc_vbrold2.json
LAMEHIP_LIB=deprecated-lame-mirror_amd/lamehip/liblamehip_prof.so python tools/stage_profile.py 1024 4 > gpurun_out/r03_stage_profile.txt 2>&1
LAMEHIP_LIB=deprecated-lame-mirror_amd/lamehip/liblamehip_prof.so python tools/stage_profile.py 1024 4 2 > gpurun_out/r03_stage_profile_vbr2.txt 2>&1
python bench.py 2>/dev/null | grep '^{"metric"' > gpurun_out/r03_bench_default.json
python bench.py --vbr 2 --no-extras 2>/dev/null | grep '^{"metric"' > gpurun_out/r03_bench_vbr2.json
python bench.py --vbr 2 --vbr-old --no-extras 2>/dev/null | grep '^{"metric"' > gpurun_out/r03_bench_vbrold2.json
ls -la gpurun_out | tail -30
cut -c1-400 gpurun_out/r03_bench_default.json
