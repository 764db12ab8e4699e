#!/bin/bash
# Round-4 measurement pass (GPU box, repo root): the bench line, the same under rocprofv3 --kernel-trace --stats, PMC traffic of the
# dominant kernel (separate FETCH_SIZE / WRITE_SIZE passes), MFMA-busy cycles per kernel of the step.  Raw files land in
# gpurun_out/r04m; the records to be judged are written straight into profiles/ (r04_*).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
python3 bench.py > $O/bench_plain.json 2> $O/bench_plain.err || { echo "plain bench failed"; tail -5 $O/bench_plain.err; }
tail -1 $O/bench_plain.json > profiles/r04_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r04 -o r04 -- python3 $R/bench.py > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || echo "rocprof bench failed"
cp $(find /tmp/prof_r04 -name "*kernel_stats.csv" | head -1) profiles/r04_kernel_stats.csv || echo "no stats csv"
tail -1 $O/bench_under_rocprof.json > profiles/r04_bench_under_rocprofv3.json
bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1 && python3 tools/pmc_postprocess.py traffic gpurun_out/pmc_traffic_raw.json profiles/r04_pmc_traffic_gemm_nt192.json "round-4 build via tools/r04_measure.sh (16-byte epilogue stores)" || echo "pmc traffic failed"
rm -rf /tmp/pmc_step4
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_step4 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_step.log 2>&1 || echo "pmc step failed"
python3 tools/pmc_postprocess.py busy $(find /tmp/pmc_step4 -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_step4 -name "*kernel_trace.csv" | head -1) 6 profiles/r04_pmc_mfma_busy_step.json "round-4 build via tools/r04_measure.sh" || echo "busy postprocess failed"
ls -la $O profiles/r04_*
