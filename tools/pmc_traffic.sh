#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters, as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), kernel-trace only; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024
# (gfx950 FETCH_SIZE counts 128-B requests as 64 B => doubled).  Run on the GPU box from the repo root.
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/tools/gemm_pmc.py > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/tools/gemm_pmc.py > $R/gpurun_out/pmc_write.log 2>&1
python3 - <<PY
import csv, glob, json, collections
def load(d, name):
    # launch order = shape order of tools/gemm_pmc.py, 4 launches per shape (the persistent kernel has one grid size for all)
    f = glob.glob(f"$R/gpurun_out/{d}/*/*counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if "gemm_nt192_kernel<0" in r["Kernel_Name"] and r["Counter_Name"] == name]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    vals = [float(r["Counter_Value"]) for r in rows]
    assert len(vals) == 16, len(vals)
    return [vals[4 * i:4 * i + 4] for i in range(4)]
fe, wr = load("pmc_fetch", "FETCH_SIZE"), load("pmc_write", "WRITE_SIZE")
M, D = 12288, 768
shapes = [(M, 3 * D, D), (M, D, 4 * D), (M, D, D), (M, D, 3 * D)]
res = []
for i, (m, n, k) in enumerate(shapes):
    res.append({"M": m, "N": n, "K": k, "fetch_kb_samples": fe[i], "write_kb_samples": wr[i]})
json.dump(res, open("$R/gpurun_out/pmc_traffic_raw.json", "w"), indent=1)
print(json.dumps(res)[:1500])
PY
