#!/bin/bash
# round 5, first GPU pass: the whole -m gpu suite on the reworked reducer / engine entry points, then the reference's own regime
# (1 and 2 clips per GPU) under the data-parallel wrapper: host enqueue time against GPU time
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_first
mkdir -p $O
python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { echo "gpu tests failed"; tail -30 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
: > $O/small_batch_dist.jsonl
for b in 1 2 8; do
  python3 bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' >> $O/small_batch_dist.jsonl
  python3 bench.py --batch $b --steps 30 --warmup 5 --force-dist --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' >> $O/small_batch_dist.jsonl
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r05_first/small_batch_dist.jsonl"):
    d = json.loads(l); print(d["config"].get("clips_per_gpu"), d["config"].get("parallelism"), d["value"], d["ms_per_step"], d.get("host_enqueue_ms_per_step"))
PY
