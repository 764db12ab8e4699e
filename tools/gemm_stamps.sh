#!/bin/bash
# Diagnostic build of the 192x192 NT GEMM with s_memtime stamps around the segments of an output tile (cdna_hip_programming.md section 7
# "In-kernel stamps"): video-tokenizer_amd/_ab/libvt_gemm_stamps.so.  usage (repo root): bash tools/gemm_stamps.sh [run]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/ab_variant.sh gemm_stamps vt_gemm192.hip "-DVT_GEMM_STAMPS" > /dev/null
[ "$1" = "run" ] && VT_HIP_LIB=$R/video-tokenizer_amd/_ab/libvt_gemm_stamps.so python3 $R/tools/gemm_stamps.py
exit 0
