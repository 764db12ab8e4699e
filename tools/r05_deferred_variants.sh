#!/bin/bash
# the deferred-epilogue GEMM: working tree against build variants (one process each, same box)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_deferred_variants.log
: > $O
for v in ${VARIANTS:-tree loader}; do
  lib=$R/video-tokenizer_amd/libvt_hip.so
  [ $v != tree ] && lib=$R/video-tokenizer_amd/_ab/libvt_$v.so
  echo "== $v" >> $O
  VT_HIP_LIB=$lib VT_BENCH_CASES=${CASES:-0123} timeout -k 10 300 python3 tools/gemm_deferred_ab.py >> $O 2>&1 || echo "(rc $?)" >> $O
done
grep -v "amdgpu.ids\|equal colsum\|: equal$" $O
