#!/bin/bash
# final-build records of round 4: smoke, the whole -m gpu suite, the bench line, and the small-batch points eager vs graph replay
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r04_final
mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { echo "smoke failed"; tail -5 $O/smoke.log; exit 1; }
python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { echo "gpu tests failed"; tail -15 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
python3 bench.py > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
cut -c1-300 $O/bench.json
: > $O/small_batch.jsonl
for b in 1 2 4; do
  python3 bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' >> $O/small_batch.jsonl
  python3 bench.py --batch $b --steps 30 --warmup 5 --graph --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' >> $O/small_batch.jsonl
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_final/small_batch.jsonl"):
    d = json.loads(l); print(d["config"].get("clips_per_gpu", d["config"]), d.get("graph"), d["value"], d["ms_per_step"])
PY
