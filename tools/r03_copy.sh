#!/bin/bash
# after `gpurun -- 'bash tools/r03_collect.sh'`: copy the summaries merged into gpurun_out/ to profiles/r03_* (kernel statistics
# cut to the first rows + the bench line of that run)
cd "$(dirname "$0")/.."
trim() { { head -8 "$1" | cut -c1-300; tail -1 "$1"; } > "$2"; }
for t in "" _vbr2 _vbrold2 _waves4; do
  trim gpurun_out/summ_r03${t}_kernel_stats.txt profiles/r03${t}_kernel_stats.txt
  cp gpurun_out/summ_r03${t}_pmc.txt profiles/r03${t}_pmc.txt
done
cp gpurun_out/summ_r03_pmc.json profiles/r03_pmc.json
cp gpurun_out/summ_r03_vbr2_pmc.json profiles/r03_pmc_vbr2.json
cp gpurun_out/summ_r03_vbrold2_pmc.json profiles/r03_pmc_vbrold2.json
cp gpurun_out/summ_r03_waves4_pmc.json profiles/r03_waves4_pmc.json
cp gpurun_out/r03_stage_profile.txt gpurun_out/r03_stage_profile_vbr2.txt gpurun_out/r03_bench_default.json gpurun_out/r03_bench_vbr2.json gpurun_out/r03_bench_vbrold2.json profiles/
