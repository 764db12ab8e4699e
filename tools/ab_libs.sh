#!/bin/bash
# interleaved timing of tools/attn_bench.py under several builds of the library: bash tools/ab_libs.sh <tag> [<tag> ...]   ("tree" = working tree)
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  for t in "$@"; do
    if [ "$t" = tree ]; then echo -n "$t: "; python $R/tools/attn_bench.py 20 2>/dev/null
    else echo -n "$t: "; VT_HIP_LIB=$R/video-tokenizer_amd/_ab/libvt_$t.so python $R/tools/attn_bench.py 20 2>/dev/null; fi
  done
done
