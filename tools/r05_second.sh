#!/bin/bash
# round 5, second GPU pass: the -m gpu suite with the measured parity figures (-rP), the bench line, the deferred-epilogue GEMM record,
# and one rocprofv3 kernel trace each of the one-clip step eager and as a graph replay (why is the replay slower?)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_second
mkdir -p $O
rm -f gpurun_out/parity_measured.jsonl
python3 -m pytest tests -x -q -m gpu -rP > $O/gpu_tests_full.log 2>&1 || { echo "gpu tests failed"; tail -30 $O/gpu_tests_full.log; exit 1; }
tail -1 $O/gpu_tests_full.log
grep "^PARITY" $O/gpu_tests_full.log > $O/parity.txt; cat $O/parity.txt | cut -c1-400
python3 bench.py > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
cut -c1-400 $O/bench.json
timeout -k 10 300 python3 tools/gemm_deferred_ab.py > $O/gemm_deferred_epilogue.log 2>&1; tail -5 $O/gemm_deferred_epilogue.log
cd /tmp && export TMPDIR=/tmp
for mode in eager graph; do
  extra=""; [ $mode = graph ] && extra="--graph"
  rocprofv3 --kernel-trace --stats -d $R/$O/prof_b1_$mode -o b1 -- python3 $R/bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline $extra > $R/$O/b1_$mode.json 2> $R/$O/b1_$mode.err || echo "rocprof $mode failed"
  f=$(find $R/$O/prof_b1_$mode -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $R/$O/b1_${mode}_kernel_stats.csv
  rm -rf $R/$O/prof_b1_$mode
done
ls $R/$O
