#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/r05_collect.sh'): every profile kept under profiles/r05_* from ONE build --
# kernel-trace statistics and PMC passes of WHAT THE DRIVER BENCHES (1024 x 60 s CBR 128, one warm-up + one timed launch), of
# BASELINE configs [2] (VBR -V2) and [4] (48 kHz CBR 320 joint stereo, 40 bursts/s) and of the old VBR loop at the extras' size,
# the LH_PROF stage profile (make -C deprecated-lame-mirror_amd/csrc prof first), then the bench lines with the fresh records in place.
set -u
cd $GRAFT_REPO_ROOT
X="--no-cpu-baseline --no-extras"
bash tools/gpu_profile.sh r05 --streams 1024 --seconds 60 --steps 1 --warmup 1 $X > gpurun_out/log_r05.txt 2>&1
bash tools/gpu_profile.sh r05_vbr2 --streams 1024 --seconds 5 --steps 2 --warmup 1 $X --vbr 2 > gpurun_out/log_r05_vbr2.txt 2>&1
bash tools/gpu_profile.sh r05_vbrold2 --streams 1024 --seconds 5 --steps 2 --warmup 1 $X --vbr 2 --vbr-old > gpurun_out/log_r05_vbrold2.txt 2>&1
bash tools/gpu_profile.sh r05_cbr320 --streams 1024 --seconds 5 --steps 2 --warmup 1 $X --samplerate 48000 --brate 320 --mode 1 --bursts 40 > gpurun_out/log_r05_cbr320.txt 2>&1
bash tools/gpu_profile.sh r05_lsf --streams 1024 --seconds 10 --steps 2 --warmup 1 $X --samplerate 22050 --brate 64 > gpurun_out/log_r05_lsf.txt 2>&1
for t in "" _vbr2 _vbrold2 _cbr320 _lsf; do
  cp gpurun_out/summ_r05${t}_pmc.json profiles/r05_pmc${t}.json
  cp gpurun_out/summ_r05${t}_pmc.txt profiles/r05${t}_pmc.txt
  cp gpurun_out/summ_r05${t}_kernel_stats.txt profiles/r05${t}_kernel_stats.txt
done
if [ -f deprecated-lame-mirror_amd/lamehip/liblamehip_prof.so ]; then
  LAMEHIP_LIB=deprecated-lame-mirror_amd/lamehip/liblamehip_prof.so python tools/stage_profile.py 1024 4 > profiles/r05_stage_profile.txt 2>&1
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
    LAMEHIP_LIB=deprecated-lame-mirror_amd/lamehip/liblamehip_prof.so python tools/stage_profile.py 1024 5 2; } > profiles/r05_vbr_stage_profile.txt 2>&1
fi
python bench.py 2>/dev/null | grep '^{"metric"' > profiles/r05_bench_default.json
python bench.py --vbr 2 --no-extras 2>/dev/null | grep '^{"metric"' > profiles/r05_bench_vbr2.json
python bench.py --vbr 2 --vbr-old --no-extras 2>/dev/null | grep '^{"metric"' > profiles/r05_bench_vbrold2.json
mkdir -p gpurun_out/profiles_r05 && cp profiles/r05* gpurun_out/profiles_r05/
cut -c1-600 profiles/r05_bench_default.json
cat profiles/r05_pmc.json
