#!/bin/bash
# Final round-3 records on the GPU box: the bench line, the same under rocprofv3 --kernel-trace --stats, PMC traffic of the dominant kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
python3 bench.py > $O/bench_plain.json 2> $O/bench_plain.err || { echo "plain bench failed"; tail -5 $O/bench_plain.err; exit 1; }
tail -1 $O/bench_plain.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r03f -o r03 -- python3 $R/bench.py > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || { echo "rocprof bench failed"; exit 1; }
cp $(find /tmp/prof_r03f -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv || echo "no stats csv"
bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1 && cp gpurun_out/pmc_traffic_raw.json $O/ || echo "pmc traffic failed"
ls -la $O
