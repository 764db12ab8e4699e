#!/bin/bash
# what the data-parallel wrapper costs on a FREE chip (world size 1, nothing resident), switch by switch: 30 steps each, one box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_wrapper_cost.jsonl
: > $O
run() { echo "# $1" >> $O; env $2 python3 bench.py $3 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'ms_per_step': d['ms_per_step'], 'clips_per_s': d['value'], 'host_enqueue_ms': d.get('host_enqueue_ms_per_step')}))" >> $O; }
run "plain (no wrapper)" "A=1" ""
run "wrapper, defaults (side stream, tail 3, one tile per workgroup, early release)" "A=1" "--force-dist"
run "wrapper, VT_WGRAD_STREAM=0" "VT_WGRAD_STREAM=0" "--force-dist"
run "wrapper, VT_WGRAD_TAIL=0" "VT_WGRAD_TAIL=0" "--force-dist"
run "wrapper, VT_DP_ONE_TILE=0" "VT_DP_ONE_TILE=0" "--force-dist"
run "wrapper, all three off (reducer only)" "VT_WGRAD_STREAM=0 VT_WGRAD_TAIL=0 VT_DP_ONE_TILE=0" "--force-dist"
run "plain again" "A=1" ""
cat $O
