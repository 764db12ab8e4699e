#!/bin/bash
# kernel statistics of the one-clip step (eager): rocprofv3 --kernel-trace --stats -- python3 bench.py --batch 1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b1 -o b1 -- python3 $R/bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/b1_bench.json 2> gpurun_out/b1_bench.err || { echo failed; tail -5 gpurun_out/b1_bench.err; exit 1; }
cp $(find /tmp/prof_b1 -name "*kernel_stats.csv" | head -1) gpurun_out/b1_kernel_stats.csv
tail -1 gpurun_out/b1_bench.json | cut -c1-200
