#!/bin/bash
# final-build records of round 5, second session: smoke, the whole -m gpu suite, the bench line, the same under rocprofv3 --kernel-trace --stats
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r05_final2
mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { echo "smoke failed"; tail -5 $O/smoke.log; exit 1; }
tail -n 1 $O/smoke.log
python3 -m pytest tests -x -q -m gpu -rP > $O/gpu_tests.log 2>&1 || { echo "gpu tests failed"; tail -n 30 $O/gpu_tests.log; exit 1; }
tail -n 1 $O/gpu_tests.log
python3 bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -n 5 $O/bench.err; exit 1; }
cut -c1-260 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_r05f
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r05f -o r05 -- python3 $R/bench.py > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || echo "rocprof bench failed"
cp $(find /tmp/prof_r05f -name "*kernel_stats.csv" | head -n 1) $O/kernel_stats.csv || echo "no stats csv"
head -n 4 $O/kernel_stats.csv | cut -c1-160
