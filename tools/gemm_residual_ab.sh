#!/bin/bash
# interleaved A/B of the fp32 + residual epilogue of the 192x192 NT kernel: one residual load per trip of the store loop (rounds 1-3,
# _ab/libvt_resloop.so = tools/ab_variant.sh resloop vt_gemm192.hip "-DVT_GEMM_RESIDUAL_IN_LOOP") vs batches of three requested one batch ahead
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  echo "== residual load inside the store loop (rounds 1-3)"; VT_HIP_LIB=$R/video-tokenizer_amd/_ab/libvt_resloop.so python3 $R/tools/gemm_bench.py 2>/dev/null | grep "f32res\|sum per"
  echo "== residual loads one batch ahead"; python3 $R/tools/gemm_bench.py 2>/dev/null | grep "f32res\|sum per"
done
