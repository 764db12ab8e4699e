#!/bin/bash
# interleaved A/B of the NT / TN GEMM shapes of the step: the library of git HEAD (_ab/libvt_base.so, tools/ab_build.sh) vs the working tree
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  echo "== HEAD"; VT_HIP_LIB=$R/video-tokenizer_amd/_ab/libvt_base.so python3 $R/tools/gemm_bench.py 2>/dev/null | grep -v "wgrad\|amdgpu"
  echo "== tree"; python3 $R/tools/gemm_bench.py 2>/dev/null | grep -v "wgrad\|amdgpu"
done
