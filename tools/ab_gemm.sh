#!/bin/bash
# interleaved A/B of the NT GEMM microbench: HEAD build vs working tree (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do
  VT_HIP_LIB=$R/video-tokenizer_amd/_ab/libvt_base.so python $R/tools/gemm_bench.py 2>/dev/null | grep -E "fwd|dgrad|sum per" | sed "s/^/base: /" | cut -c1-150
  python $R/tools/gemm_bench.py 2>/dev/null | grep -E "fwd|dgrad|sum per" | sed "s/^/new:  /" | cut -c1-150
done
