#!/bin/bash
# which bf16 epilogue outputs should leave by streaming (nt) stores?  -DVT_GEMM_PLAIN_STORES variants of vt_gemm192.hip (tools/ab_sched_build.sh), whole 8-clip step, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_store_kind
mkdir -p $O
for i in 1 2 3; do
  for t in st0 st_g st_bf st_dg st_all st_bf_g_dg; do
    echo -n "$t: " >> $O/step.log
    VT_HIP_LIB=$R/video-tokenizer_amd/_ab/libvt_$t.so python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'])" >> $O/step.log
  done
done
cat $O/step.log
