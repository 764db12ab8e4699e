#!/bin/bash
# A/B timing helper: build the library of git HEAD into video-tokenizer_amd/_ab/libvt_base.so (git-ignored, travels to the
# GPU box) next to the working-tree build; select it with VT_HIP_LIB=<path>.  Run in the dev container from the repo root.
set -e
R=$(git rev-parse --show-toplevel)
T=$(mktemp -d)
mkdir -p $T/video-tokenizer_amd/csrc $T/include $R/video-tokenizer_amd/_ab
git -C $R archive HEAD video-tokenizer_amd/csrc include | tar -x -C $T
objs=""
for f in $T/video-tokenizer_amd/csrc/*.hip $T/video-tokenizer_amd/csrc/*.cpp; do
  o=$T/$(basename $f).o
  x=""; case $f in *.cpp) x="-x hip";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w $x -c $f -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/video-tokenizer_amd/_ab/libvt_base.so $objs
rm -rf $T
ls -la $R/video-tokenizer_amd/_ab/libvt_base.so
