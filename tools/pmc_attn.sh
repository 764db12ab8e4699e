#!/bin/bash
# PMC passes over the attention kernels only (tools/attn_bench.py 3): issue / wait / co-execution / LDS counters.
# usage: bash tools/pmc_attn.sh <tag>   GPU box, repo root.  Output keyed per kernel (attn_fwd_kernel<64, false>, attn_bwd_dq_kernel<...>, attn_bwd_dkv_kernel<...>).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-attn}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE" \
           "SQ_BUSY_CU_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG/p$i -- python3 $R/tools/attn_bench.py 3 > $R/gpurun_out/pmc_${TAG}_p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections, json, re
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_$TAG/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn" not in k: continue
        m = re.search(r"(attn_\w+(<[^>]*>)?)", k)     # round 4: keyed per kernel (the old split("(") key merged fwd / dQ / dK-dV into "void ")
        res[m.group(1) if m else k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items()}
json.dump(out, open("$R/gpurun_out/pmc_$TAG.json", "w"), indent=1)
for k, d in out.items():
    print(k)
    for c, v in sorted(d.items()): print(f"   {c:32s} {v:16.0f}")
PY
