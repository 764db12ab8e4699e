#!/bin/bash
# Variant builds of ONE source file under different LLVM scheduler options, each linked with the tree's other objects into
# video-tokenizer_amd/_ab/libvt_<tag>.so (selected by VT_HIP_LIB, see tools/ab_libs.sh).  Prints registers / spills per kernel.
#   bash tools/ab_sched_build.sh vt_attention.hip maxilp "-mllvm -amdgpu-sched-strategy=max-ilp" [tag "flags" ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
P=$R/video-tokenizer_amd
SRC=$1; shift
BASE=$(basename $SRC .hip)
mkdir -p $P/_ab
while [ $# -ge 2 ]; do
  TAG=$1; FL=$2; shift 2
  O=$P/_ab/${BASE}_$TAG.o
  if ! /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $FL -c $P/csrc/$SRC -o $O -Rpass-analysis=kernel-resource-usage > $P/_ab/${BASE}_$TAG.log 2>&1; then
    echo "$TAG: COMPILE FAILED"; tail -3 $P/_ab/${BASE}_$TAG.log; continue
  fi
  OBJS=""
  for f in $P/_obj/*.o; do [ "$(basename $f)" = "$BASE.o" ] && OBJS="$OBJS $O" || OBJS="$OBJS $f"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/_ab/libvt_$TAG.so $OBJS
  SP=$(grep -c "VGPRs Spill: [1-9]\|ScratchSize \[bytes/lane\]: [1-9]" $P/_ab/${BASE}_$TAG.log)
  echo "$TAG: built, spill lines $SP"
  python3 - $P/_ab/${BASE}_$TAG.log <<'PY'
import re, sys
name = None
for ln in open(sys.argv[1]):
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        name = m.group(1)
    m = re.search(r" VGPRs: (\d+)", ln)
    if m and name and ("Lb0" in name or "gemm_nt192" in name or "ln_" in name) and "ILi32" not in name:
        print("   ", name[:70], "VGPRs", m.group(1))
PY
done
