#!/bin/bash
# LLVM scheduler options on the compiler-scheduled kernels (tools/ab_sched_build.sh builds the variants): attention micro-benchmark and whole step, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_sched
mkdir -p $O
bash tools/ab_libs.sh base maxilp maxmem itermaxocc trackers nopost nounclust > $O/attn.log 2>&1
cat $O/attn.log
for i in 1 2; do
  for t in base g_maxilp g_maxmem g_itermaxocc g_trackers g_nopost; do
    echo -n "$t: " >> $O/step.log
    VT_HIP_LIB=$R/video-tokenizer_amd/_ab/libvt_$t.so python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'])" >> $O/step.log
  done
done
cat $O/step.log
