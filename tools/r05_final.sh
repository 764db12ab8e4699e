#!/bin/bash
# final-build records of round 5: smoke, the whole -m gpu suite (with the measured parity figures), the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_final
mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { echo "smoke failed"; tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
python3 -m pytest tests -x -q -m gpu -rP > $O/gpu_tests.log 2>&1 || { echo "gpu tests failed"; tail -30 $O/gpu_tests.log; exit 1; }
tail -1 $O/gpu_tests.log
python3 bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
cut -c1-260 $O/bench.json
python3 bench.py --force-dist --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' > $O/bench_force_dist.json; cut -c1-200 $O/bench_force_dist.json
