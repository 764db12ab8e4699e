#!/bin/bash
# idle time between consecutive kernels of the 8-clip step on the GPU's own clock: kernel trace of the REPLAYED step (the host is out of the picture),
# next to the un-traced eager and replayed step times
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r05_gaps
mkdir -p $O
for m in "" "--graph"; do
  echo -n "bench $m: " >> $O/step.log
  python3 bench.py $m --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'], r.get('host_enqueue_ms_per_step'))" >> $O/step.log
done
cat $O/step.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --graph --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $O/trace.log 2>&1
F=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/graph_gaps.py $F 4000 > $O/gaps.log 2>&1
cat $O/gaps.log
rm -rf $O/trace
