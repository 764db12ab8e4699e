#!/bin/bash
# interleaved A/B of the whole training step: HEAD build (tools/ab_build.sh) vs working tree (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do
  VT_HIP_LIB=$R/video-tokenizer_amd/_ab/libvt_base.so python $R/bench.py --no-cpu-baseline --no-roofline --steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base step', d['value'], d['ms_per_step'])"
  python $R/bench.py --no-cpu-baseline --no-roofline --steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new  step', d['value'], d['ms_per_step'])"
done
