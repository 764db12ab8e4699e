#!/bin/bash
# HBM-side traffic of EVERY kernel of the training step: FETCH_SIZE and WRITE_SIZE in separate passes over
# `bench.py --steps 2 --warmup 1`, bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 correction of
# /opt/skills/guides/MI355X_MICROARCH.md), aggregated per kernel name and per step.  GPU box, from the repo root.
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_step_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_step_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_step_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_step_write.log 2>&1
python3 - <<PY
import csv, glob, json, collections, re
def load(d, name):
    f = glob.glob(f"$R/gpurun_out/{d}/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:70]
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    return agg
fe, wr = load("pmc_step_fetch", "FETCH_SIZE"), load("pmc_step_write", "WRITE_SIZE")
steps = 3.0   # 1 warm-up + 2 timed steps, identical work
rows = []
for k in sorted(set(fe) | set(wr), key=lambda k: -(2 * fe[k][0] + wr[k][0])):
    rd, wrb = 2 * fe[k][0] * 1024 / steps, wr[k][0] * 1024 / steps
    rows.append({"kernel": k, "launches_per_step": round(max(fe[k][1], wr[k][1]) / steps, 1), "read_MB_per_step": round(rd / 1e6, 1),
                 "write_MB_per_step": round(wrb / 1e6, 1)})
tot_r, tot_w = sum(r["read_MB_per_step"] for r in rows), sum(r["write_MB_per_step"] for r in rows)
out = {"what": "HBM-side bytes per training step (config B, 8 clips) per kernel: (2*FETCH_SIZE, WRITE_SIZE)*1024, separate rocprofv3 --pmc passes over bench.py --steps 2 --warmup 1",
       "total_read_GB_per_step": round(tot_r / 1e3, 2), "total_write_GB_per_step": round(tot_w / 1e3, 2), "kernels": rows}
json.dump(out, open("$R/gpurun_out/pmc_step_traffic.json", "w"), indent=1)
print("total read GB", out["total_read_GB_per_step"], "write GB", out["total_write_GB_per_step"])
for r in rows[:25]: print(r)
PY
