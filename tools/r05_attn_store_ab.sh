#!/bin/bash
# guide T21 on the attention kernels' output stores: 8-byte row-per-lane stores (-DVT_ATTN_STORE8=1) against permlane32_swap + 16-byte stores; micro-benchmark and whole step, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_attn_store
mkdir -p $O
python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention or attn" > $O/tests.log 2>&1; tail -n 2 $O/tests.log
bash tools/ab_libs.sh at8 at16 at8 at16 > $O/attn.log 2>&1
cat $O/attn.log
for i in 1 2 3; do
  for t in at8 at16; do
    echo -n "$t: " >> $O/step.log
    VT_HIP_LIB=$R/video-tokenizer_amd/_ab/libvt_$t.so python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'])" >> $O/step.log
  done
done
cat $O/step.log
