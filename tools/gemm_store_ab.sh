#!/bin/bash
# interleaved A/B of the NT GEMM epilogue store width: _ab/libvt_store8.so (8-byte stores, rounds 1-3) vs the working tree (16-byte)
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  echo "== store8"; VT_HIP_LIB=$R/video-tokenizer_amd/_ab/libvt_store8.so python $R/tools/gemm_bench.py 2>/dev/null | grep -v "wgrad\|amdgpu"
  echo "== store16"; python $R/tools/gemm_bench.py 2>/dev/null | grep -v "wgrad\|amdgpu"
done
