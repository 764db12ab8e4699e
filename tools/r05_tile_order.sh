#!/bin/bash
# Records for the column-block tile order of the 192x192 NT kernel (round 5): the GEMM tests, the bench line, PMC traffic of the dominant kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05t
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm" > $O/gemm_tests.log 2>&1 || { echo "gemm tests failed"; tail -30 $O/gemm_tests.log; exit 1; }
tail -1 $O/gemm_tests.log
python3 bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
cut -c1-300 $O/bench.json
bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1 && python3 tools/pmc_postprocess.py traffic gpurun_out/pmc_traffic_raw.json $O/r05_pmc_traffic_gemm_nt192_tile_order.json "round-5 build with the column-block tile order via tools/r05_tile_order.sh" || echo "pmc traffic failed"
grep -o '"name": "[^"]*"\|"ratio": [0-9.]*' $O/r05_pmc_traffic_gemm_nt192_tile_order.json | paste - - 
