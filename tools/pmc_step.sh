#!/bin/bash
# MFMA-pipe utilisation of every kernel of the training step (SQ_VALU_MFMA_BUSY_CYCLES) and HBM-side traffic of the
# codebook search (FETCH_SIZE / WRITE_SIZE in separate passes).  Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_step -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_step.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_vq_fetch -- python3 $R/tools/vq_pmc.py > $R/gpurun_out/pmc_vq_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_vq_write -- python3 $R/tools/vq_pmc.py > $R/gpurun_out/pmc_vq_write.log 2>&1
echo pmc_step done
