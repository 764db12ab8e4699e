#!/bin/bash
# Round-2 measurement pass (GPU box, repo root): the bench line under rocprofv3 --kernel-trace --stats, PMC traffic of the dominant
# kernel (separate FETCH_SIZE / WRITE_SIZE passes), MFMA-busy cycles per kernel of the step, the 'sq' search, our GEMMs vs hipBLASLt.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
python3 bench.py > $O/bench_plain.json 2> $O/bench_plain.err || echo "plain bench failed"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r02 -o r02 -- python3 $R/bench.py > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || echo "rocprof bench failed"
cp $(find /tmp/prof_r02 -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv || echo "no stats csv"
bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1 && cp gpurun_out/pmc_traffic_raw.json $O/ || echo "pmc traffic failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_step -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_step.log 2>&1 || echo "pmc step failed"
cp $(find /tmp/pmc_step -name "*counter_collection.csv" | head -1) $O/pmc_step_counters.csv || true
cp $(find /tmp/pmc_step -name "*kernel_trace.csv" | head -1) $O/pmc_step_trace.csv || true
python3 tools/vq_pmc.py sq > $O/sq_search.log 2>&1 || echo "sq timing failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_sq_f -- python3 $R/tools/vq_pmc.py sq > /dev/null 2>&1 && cp $(find /tmp/pmc_sq_f -name "*counter_collection.csv" | head -1) $O/sq_fetch.csv
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_sq_w -- python3 $R/tools/vq_pmc.py sq > /dev/null 2>&1 && cp $(find /tmp/pmc_sq_w -name "*counter_collection.csv" | head -1) $O/sq_write.csv
python3 tools/gemm_vs_hipblaslt.py > $O/gemm_vs_hipblaslt.log 2>&1 || echo "hipblaslt comparison failed"
ls -la $O
