#!/bin/bash
# Build an A/B variant of ONE source file with extra -D flags into video-tokenizer_amd/_ab/libvt_<tag>.so (the other objects are the
# working tree's).  usage: bash tools/ab_variant.sh <tag> <file.hip> "<-DFLAG ...>"   then  VT_HIP_LIB=.../_ab/libvt_<tag>.so python tools/...
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; SRC=$2; FLAGS=$3
cd $R/video-tokenizer_amd
mkdir -p _ab
base=$(basename $SRC .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $FLAGS -c csrc/$SRC -o _ab/${base}_$TAG.o
objs=$(ls _obj/*.o | grep -v "/${base}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/libvt_$TAG.so $objs _ab/${base}_$TAG.o
ls -la _ab/libvt_$TAG.so
