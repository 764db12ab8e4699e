#!/bin/bash
# second half of tools/r04_measure.sh on its own (PMC passes only): whole-step MFMA busy per kernel, counters of the grouped weight-gradient GEMM
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04m
mkdir -p $O/profiles
cd /tmp && export TMPDIR=/tmp
cd $R
rm -rf /tmp/pmc_step4
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_step4 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_step.log 2>&1 || echo "pmc step failed"
python3 tools/pmc_postprocess.py busy $(find /tmp/pmc_step4 -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_step4 -name "*kernel_trace.csv" | head -1) 6 $O/profiles/r04_pmc_mfma_busy_step.json "round-4 build via tools/r04_measure_pmc.sh" > /dev/null || echo "busy postprocess failed"
# counters of the grouped weight-gradient GEMM (verdict item 4): matrix-pipe busy, LDS conflicts / activity, waits, clock; HBM-side fetch
rm -rf /tmp/pmc_tn_a /tmp/pmc_tn_b /tmp/pmc_tn_c
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc_tn_a -- python3 $R/tools/gemm_tn_ab.py 3 > $O/pmc_tn_a.log 2>&1 || echo "pmc tn a failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_tn_b -- python3 $R/tools/gemm_tn_ab.py 3 > $O/pmc_tn_b.log 2>&1 || echo "pmc tn b failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_tn_c -- python3 $R/tools/gemm_tn_ab.py 3 > $O/pmc_tn_c.log 2>&1 || echo "pmc tn c failed"
python3 - <<PY
import csv, glob, json, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in ("/tmp/pmc_tn_a", "/tmp/pmc_tn_b", "/tmp/pmc_tn_c"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm_tn192" not in k or int(r["Grid_Size"]) < 768 * 512: continue
            out["pipelined" if "tn192p" in k else "burst (round 1)"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm_tn192" in k and int(r["Grid_Size_X"]) >= 768 * 512:
                dur["pipelined" if "tn192p" in k else "burst (round 1)"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
res = {"what": "grouped weight-gradient GEMM, the step's 4-block launch (768 tiles, 1.2 GB of operands, tools/gemm_tn_ab.py under rocprofv3 --pmc, three passes); averages per launch",
       "kernels": {}}
for k, c in out.items():
    t = sum(dur[k]) / max(len(dur[k]), 1)
    e = {n: sum(v) / len(v) for n, v in c.items()}
    e["duration_us_under_profiler"] = round(t / 1e3, 1)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e: e["mfma_pipe_busy_at_2.1GHz"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * t * 2.1), 4)
    if "GRBM_GUI_ACTIVE" in e: e["clock_GHz_from_GRBM_GUI_ACTIVE"] = round(e["GRBM_GUI_ACTIVE"] / 8 / t, 3)
    if "FETCH_SIZE" in e: e["hbm_side_fetch_bytes (2 x FETCH_SIZE x 1024)"] = 2 * e["FETCH_SIZE"] * 1024
    res["kernels"][k] = e
json.dump(res, open("$R/gpurun_out/r04m/profiles/r04_pmc_gemm_tn192.json", "w"), indent=1)
print(json.dumps(res)[:1200])
PY
ls -la $O/profiles
