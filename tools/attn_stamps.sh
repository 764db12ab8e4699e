#!/bin/bash
# Diagnostic build of the three attention kernels with s_memtime stamps around the segments of a tile iteration
# (cdna_hip_programming.md section 7 "In-kernel stamps"): builds video-tokenizer_amd/_ab/libvt_attn_stamps.so next to the real library.
# Read the SHARES it prints, never its run time.  usage (repo root, here or on the GPU box): bash tools/attn_stamps.sh [run]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/video-tokenizer_amd
mkdir -p _ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DVT_ATTN_STAMPS -c csrc/vt_attention.hip -o _ab/vt_attention_stamps.o
objs=$(ls _obj/*.o | grep -v "vt_attention.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/libvt_attn_stamps.so $objs _ab/vt_attention_stamps.o
[ "$1" = "run" ] && VT_HIP_LIB=$R/video-tokenizer_amd/_ab/libvt_attn_stamps.so python3 $R/tools/attn_stamps.py
exit 0
