#!/bin/bash
# interleaved A/B of the attention kernels at the step's shape: HEAD build (tools/ab_build.sh) vs working tree (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  echo -n "base: "; VT_HIP_LIB=$R/video-tokenizer_amd/_ab/libvt_base.so python $R/tools/attn_bench.py 20
  echo -n "new:  "; python $R/tools/attn_bench.py 20
done
