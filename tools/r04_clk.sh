#!/bin/bash
# GPU box: sample the engine clock while the default bench workload runs (is the kernel clock-limited?)
cd $GRAFT_REPO_ROOT
python bench.py --streams 1024 --seconds 20 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r04_base.log 2>&1 &
BP=$!
: > gpurun_out/r04_clk.log
while kill -0 $BP 2>/dev/null; do
  echo "t=$(date +%s.%N)" >> gpurun_out/r04_clk.log
  rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|Power (W)" >> gpurun_out/r04_clk.log
  sleep 1
done
wait $BP
tail -1 gpurun_out/r04_base.log | cut -c1-600
grep sclk gpurun_out/r04_clk.log | sort | uniq -c | sort -rn | head
grep "Power (W)" gpurun_out/r04_clk.log | sort | uniq -c | sort -rn | head -5
