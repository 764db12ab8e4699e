#!/bin/bash
# MFMA-pipe busy cycles per kernel of the step on the second session's final build (attention stores widened)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_pmc2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_step6
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_step6 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_step.log 2>&1 || echo "pmc step failed"
cd $R
python3 tools/pmc_postprocess.py busy $(find /tmp/pmc_step6 -name "*counter_collection.csv" | head -n 1) $(find /tmp/pmc_step6 -name "*kernel_trace.csv" | head -n 1) 6 $O/r05_pmc_mfma_busy_step_second_session.json "second session's final build via tools/r05_pmc_busy2.sh" | tail -n 20
