#!/bin/bash
# Samples power, clocks and the power cap (rocm-smi) every 0.25 s while bench.py runs: does the step run at the power limit?
# usage (GPU box, repo root): bash tools/power_clock_trace.sh > gpurun_out/power_trace.log
rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -v "^$" | head -30
echo "---- sampling during bench.py --steps 800"
python bench.py --steps 800 --warmup 5 --no-cpu-baseline --no-roofline > /tmp/bench_pw.json 2>/dev/null &
BP=$!
sleep 5
for i in $(seq 1 60); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '
  echo
  sleep 0.25
  kill -0 $BP 2>/dev/null || break
done
wait $BP
cat /tmp/bench_pw.json | cut -c1-200
