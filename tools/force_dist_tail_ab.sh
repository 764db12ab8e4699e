#!/bin/bash
# bench.py --force-dist (RCCL group of one rank, bucketed reducer) with the single-GPU launch schedule (VT_WGRAD_TAIL=0) and the data-parallel one (3), interleaved
for t in 0 3 0 3; do
  echo -n "VT_WGRAD_TAIL=$t: "; VT_WGRAD_TAIL=$t python3 bench.py --force-dist --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
