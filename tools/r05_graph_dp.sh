#!/bin/bash
# the reference's own regime (1 / 2 clips per GPU) under the data-parallel wrapper at world size 1: eager stage-by-stage backward vs the captured step
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_graph_dp
mkdir -p $O
for b in 1 2 8; do
  for m in "" "--graph"; do
    python3 bench.py --batch $b --force-dist $m --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python3 -c "
import sys,json
r=json.loads(sys.stdin.read())
print(json.dumps({'clips_per_gpu': $b, 'mode': 'wrapper' + (' + graph replay' if '$m' else ' eager'), 'ms_per_step': r['ms_per_step'], 'clips_per_s': r['value'], 'host_enqueue_ms_per_step': r.get('host_enqueue_ms_per_step')}))" >> $O/small_batch_wrapper_eager_vs_graph.jsonl
  done
done
cat $O/small_batch_wrapper_eager_vs_graph.jsonl
