#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/r06_collect.sh'): every profile kept under profiles/r06_* from ONE build --
# kernel-trace statistics and PMC passes of WHAT THE DRIVER BENCHES (1024 x 60 s CBR 128, one warm-up + one timed launch), of
# BASELINE configs [2] (VBR -V2) and [4] (48 kHz CBR 320 joint stereo, 40 bursts/s) and of the old VBR loop at the extras' size,
# the LH_PROF stage profile, the search trace and the analysis phase profile (make -C deprecated-lame-mirror_amd/csrc prof trace aprof first), then the bench lines with the fresh records in place.
set -u
cd $GRAFT_REPO_ROOT
X="--no-cpu-baseline --no-extras"
bash tools/gpu_profile.sh r06 --streams 1024 --seconds 60 --steps 1 --warmup 1 $X > gpurun_out/log_r06.txt 2>&1
bash tools/gpu_profile.sh r06_vbr2 --streams 1024 --seconds 5 --steps 2 --warmup 1 $X --vbr 2 > gpurun_out/log_r06_vbr2.txt 2>&1
bash tools/gpu_profile.sh r06_vbrold2 --streams 1024 --seconds 5 --steps 2 --warmup 1 $X --vbr 2 --vbr-old > gpurun_out/log_r06_vbrold2.txt 2>&1
bash tools/gpu_profile.sh r06_cbr320 --streams 1024 --seconds 5 --steps 2 --warmup 1 $X --samplerate 48000 --brate 320 --mode 1 --bursts 40 > gpurun_out/log_r06_cbr320.txt 2>&1
bash tools/gpu_profile.sh r06_lsf --streams 1024 --seconds 10 --steps 2 --warmup 1 $X --samplerate 22050 --brate 64 > gpurun_out/log_r06_lsf.txt 2>&1
for t in "" _vbr2 _vbrold2 _cbr320 _lsf; do
  cp gpurun_out/summ_r06${t}_pmc.json profiles/r06_pmc${t}.json
  cp gpurun_out/summ_r06${t}_pmc.txt profiles/r06${t}_pmc.txt
  cp gpurun_out/summ_r06${t}_kernel_stats.txt profiles/r06${t}_kernel_stats.txt
done
if [ -f deprecated-lame-mirror_amd/lamehip/liblamehip_prof.so ]; then
  LAMEHIP_LIB=deprecated-lame-mirror_amd/lamehip/liblamehip_prof.so python tools/stage_profile.py 1024 4 > profiles/r06_stage_profile.txt 2>&1
  { cat <<'HDR'
# LH_PROF build, VBR -V2 (vbr_mtrh), 1024 x 5 s: LAMEHIP_LIB=.../liblamehip_prof.so python tools/stage_profile.py 1024 5 2
# With a VBR quality the slots mean: "outer_loop" = geometry + scalefactor search, "count_bits total" = lh_vbr_noisy_n (16 calls
# per frame and wave: "calc_noise calls"), "quantise part" = its phase A (error sums of the groups of four lines at up to three
# trial steps), "calc_noise" = quantise + count of the final steps, "bin_search" = geometry, "balance_noise" = constrain + bitcount.
# BEFORE commit e2a09ba (the band sums as 64-lane ds_add_f32 to the bands' words; same command, same box class):
#    frame total 410 315   outer_loop 199 556   count_bits total 172 497   quantise part 155 506   kernel 44.40 ms (with marks)
# Experiments on that build (results wrong, times telling): every look-up to address 0: quantise part 102 412; band sums stored
# instead of added: 76 213 (kernel 33.35 ms); on this build with every look-up issued twice: 141 792 (from 92 200).
HDR
    LAMEHIP_LIB=deprecated-lame-mirror_amd/lamehip/liblamehip_prof.so python tools/stage_profile.py 1024 5 2; } > profiles/r06_vbr_stage_profile.txt 2>&1
fi
# the search's segments (make -C deprecated-lame-mirror_amd/csrc trace) and the analysis kernels' phases (make aprof)
[ -f deprecated-lame-mirror_amd/lamehip/liblamehip_trace.so ] && LAMEHIP_LIB=$PWD/deprecated-lame-mirror_amd/lamehip/liblamehip_trace.so python tools/trace_profile.py 1024 4 > profiles/r06_trace_profile.txt 2>&1
[ -f deprecated-lame-mirror_amd/lamehip/liblamehip_aprof.so ] && LAMEHIP_LIB=$PWD/deprecated-lame-mirror_amd/lamehip/liblamehip_aprof.so python tools/an_profile.py 1024 4 > profiles/r06_an_profile.txt 2>&1
python bench.py 2>/dev/null | grep '^{"metric"' > profiles/r06_bench_default.json
python bench.py --vbr 2 --no-extras 2>/dev/null | grep '^{"metric"' > profiles/r06_bench_vbr2.json
python bench.py --vbr 2 --vbr-old --no-extras 2>/dev/null | grep '^{"metric"' > profiles/r06_bench_vbrold2.json
mkdir -p gpurun_out/profiles_r06 && cp profiles/r06* gpurun_out/profiles_r06/
cut -c1-600 profiles/r06_bench_default.json
cat profiles/r06_pmc.json
