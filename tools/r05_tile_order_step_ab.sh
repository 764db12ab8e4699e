#!/bin/bash
# Whole-step A/B of the NT kernels' tile order on ONE box: row-major list (VT_GEMM_TILE_ORDER=0) against the automatic column blocks, interleaved;
# at 8 clips (the 192x192 kernel's wide launches) and at 1 and 2 clips (the 128x128 kernel: the reference's own one-clip-per-GPU regime)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05t
mkdir -p $O
cd $R
: > $O/step_ab.jsonl
for batch in 8 1 2; do
  for rep in 1 2 3; do
    for ord in 0 auto; do
      if [ $ord = auto ]; then unset VT_GEMM_TILE_ORDER; else export VT_GEMM_TILE_ORDER=$ord; fi
      python3 bench.py --batch $batch --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'clips':$batch,'order':'$ord','rep':$rep,'clips_s':d['value'],'ms_per_step':d['ms_per_step']}))" >> $O/step_ab.jsonl
    done
  done
done
cat $O/step_ab.jsonl
