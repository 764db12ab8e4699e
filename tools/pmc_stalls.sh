#!/bin/bash
# Where do the waves of the GEMM and attention kernels spend their cycles?  Issue / wait counters in separate passes
# (one rocprofv3 --pmc run each; kernel-trace only).  Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_stalls/p$i -- python3 $R/tools/stall_workload.py > $R/gpurun_out/pmc_stalls_p$i.log 2>&1 || echo "pass $i ($set) failed"
done
echo pmc_stalls done
