#!/bin/bash
# Round-5 measurement pass (GPU box, repo root): the bench line, the same under rocprofv3 --kernel-trace --stats, PMC traffic of the dominant
# kernel (separate FETCH_SIZE / WRITE_SIZE passes), MFMA-busy cycles per kernel of the step, the codebook search's traffic, and the one-clip
# step eager vs graph replay under the kernel tracer.  Everything lands in gpurun_out/r05m (copied into profiles/ in the dev container).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
python3 bench.py > $O/bench_plain.json 2> $O/bench_plain.err || { echo "plain bench failed"; tail -5 $O/bench_plain.err; }
tail -1 $O/bench_plain.json > $O/r05_bench.json
rm -rf /tmp/prof_r05
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r05 -o r05 -- python3 $R/bench.py > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || echo "rocprof bench failed"
cp $(find /tmp/prof_r05 -name "*kernel_stats.csv" | head -1) $O/r05_kernel_stats.csv || echo "no stats csv"
tail -1 $O/bench_under_rocprof.json > $O/r05_bench_under_rocprofv3.json
bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1 && python3 tools/pmc_postprocess.py traffic gpurun_out/pmc_traffic_raw.json $O/r05_pmc_traffic_gemm_nt192.json "round-5 build via tools/r05_measure.sh" || echo "pmc traffic failed"
rm -rf /tmp/pmc_step5
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_step5 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_step.log 2>&1 || echo "pmc step failed"
python3 tools/pmc_postprocess.py busy $(find /tmp/pmc_step5 -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_step5 -name "*kernel_trace.csv" | head -1) 6 $O/r05_pmc_mfma_busy_step.json "round-5 build via tools/r05_measure.sh" > /dev/null || echo "busy postprocess failed"
# the codebook search (north star: "achieved HBM GB/s on the codebook argmin"): HBM-side bytes per launch, two passes
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_vq_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_vq_$c -- python3 $R/tools/vq_pmc.py > $O/pmc_vq_$c.log 2>&1 || echo "pmc vq $c failed"
done
python3 - <<PY
import csv, glob, json, collections
res = {"what": "vq_search_kernel at N = K = 8192, d = 24 (tools/vq_pmc.py under rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate passes); per launch, by kernel instantiation (index mode)",
       "bytes": "hbm_side = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes); algorithmic = 4 N d + 4 K d + 8 N + 4 N d = 2.4 MB", "kernels": {}}
vals = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/pmc_vq_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "vq_search_kernel" in r["Kernel_Name"]:
                vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(f"/tmp/pmc_vq_{c}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "vq_search_kernel" in r["Kernel_Name"]:
                dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, c in vals.items():
    t = sum(dur[k]) / max(len(dur[k]), 1) / 1e3
    fe = sum(c.get("FETCH_SIZE", [0])) / max(len(c.get("FETCH_SIZE", [1])), 1)
    wr = sum(c.get("WRITE_SIZE", [0])) / max(len(c.get("WRITE_SIZE", [1])), 1)
    b = (2 * fe + wr) * 1024
    res["kernels"][k] = {"us_under_profiler": round(t, 1), "hbm_side_bytes": int(b), "hbm_side_GBps": round(b / t / 1e3, 1), "exact_fp32_TFLOPs": round(2 * 8192 * 8192 * 24 / t / 1e6, 1)}
json.dump(res, open("$O/r05_pmc_vq_search.json", "w"), indent=1)
print(json.dumps(res)[:900])
PY
# the one-clip step under the kernel tracer, eager and as a graph replay (why is the replay slower on the GPU?)
for mode in eager graph; do
  extra=""; [ $mode = graph ] && extra="--graph"
  rm -rf /tmp/prof_b1_$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b1_$mode -o b1 -- python3 $R/bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline $extra > $O/b1_$mode.json 2> $O/b1_$mode.err || echo "rocprof $mode failed"
  cp $(find /tmp/prof_b1_$mode -name "*kernel_stats.csv" | head -1) $O/r05_b1_${mode}_kernel_stats.csv || echo "no b1 $mode stats"
done
ls -la $O
