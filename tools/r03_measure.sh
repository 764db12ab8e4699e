#!/bin/bash
# Round-3 measurement pass (GPU box, repo root): the bench line, the same under rocprofv3 --kernel-trace --stats, PMC traffic of the
# dominant kernel (separate FETCH_SIZE / WRITE_SIZE passes), MFMA-busy cycles per kernel of the step, the 1 / 2 / 4-clip points eager
# and graph-replayed, the attention backward A/B.  Everything lands in gpurun_out/r03m; copy what is to be judged into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
python3 bench.py > $O/bench_plain.json 2> $O/bench_plain.err || echo "plain bench failed"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r03 -o r03 -- python3 $R/bench.py > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || echo "rocprof bench failed"
cp $(find /tmp/prof_r03 -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv || echo "no stats csv"
bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1 && cp gpurun_out/pmc_traffic_raw.json $O/ || echo "pmc traffic failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_step3 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_step.log 2>&1 || echo "pmc step failed"
cp $(find /tmp/pmc_step3 -name "*counter_collection.csv" | head -1) $O/pmc_step_counters.csv || true
cp $(find /tmp/pmc_step3 -name "*kernel_trace.csv" | head -1) $O/pmc_step_trace.csv || true
for b in 1 2 4; do
  for g in "" "--graph"; do
    python3 bench.py --batch $b --steps 30 --warmup 5 --optimizer fused --no-cpu-baseline --no-roofline $g > $O/bench_b${b}${g}.json 2> $O/bench_b${b}${g}.err || echo "bench b$b $g failed"
  done
done
# (round 4: tools/attn_bwd_ab.py left the tree with vt_attention_bwd_fused; its records are profiles/r03_attention_bwd_*)
ls -la $O
