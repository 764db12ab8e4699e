"""Turn the raw rocprofv3 outputs of tools/r04_measure.sh into the records committed under profiles/ (the files bench.py reads):
  traffic  <pmc_traffic_raw.json>            -> r0N_pmc_traffic_gemm_nt192.json   (2*FETCH_SIZE + WRITE_SIZE, per launch, vs algorithmic bytes)
  busy     <counter_collection.csv> <kernel_trace.csv> <steps traced> -> r0N_pmc_mfma_busy_step.json (SQ_VALU_MFMA_BUSY_CYCLES per kernel and step)
usage: python tools/pmc_postprocess.py traffic raw.json out.json "<note>" | busy counters.csv trace.csv nsteps out.json "<note>" """
import collections
import csv
import json
import sys


def traffic(raw, out, note):
    rows = json.load(open(raw))
    names = ["qkv fwd", "fc1 dgrad", "proj dgrad", "qkv dgrad"]
    per, tb, ab = [], [], []
    for name, r in zip(names, rows):
        fe = sum(r["fetch_kb_samples"][1:]) / len(r["fetch_kb_samples"][1:])      # first launch of a shape: cold caches
        wr = sum(r["write_kb_samples"][1:]) / len(r["write_kb_samples"][1:])
        t = (2 * fe + wr) * 1024
        a = 2 * (r["M"] * r["K"] + r["N"] * r["K"] + r["M"] * r["N"])
        per.append({"name": name, "M": r["M"], "N": r["N"], "K": r["K"], "FETCH_SIZE_KB": round(fe, 1), "WRITE_SIZE_KB": round(wr, 1),
                    "traffic_bytes": t, "algorithmic_bytes": a, "ratio": round(t / a, 3)})
        tb.append(t)
        ab.append(a)
    json.dump({"kernel": "gemm_nt192_kernel<VT_EPI_BF16, 4>", "tool": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), tools/pmc_traffic.sh; " + note,
               "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE tallies 128-B requests as 64 B (MI355X_MICROARCH.md section HBM); counters are L2 fabric requests, Infinity-Cache hits included",
               "per_shape": per, "traffic_bytes_per_launch_mean": sum(tb) / len(tb), "algorithmic_bytes_per_launch_mean": sum(ab) / len(ab),
               "ratio_mean": round(sum(tb) / sum(ab), 3)}, open(out, "w"), indent=1)
    print(open(out).read()[:600])


def busy(counters, trace, nsteps, out, note, ghz=2.1):
    dur = {}
    for r in csv.DictReader(open(trace)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for r in csv.DictReader(open(counters)):
        if r["Counter_Name"] != "SQ_VALU_MFMA_BUSY_CYCLES":
            continue
        d = dur.get(r["Dispatch_Id"])
        if d is None:
            continue
        k = d[1].replace("(anonymous namespace)::", "")
        k = (k[5:] if k.startswith("void ") else k).split("(")[0][-64:]
        a = agg[k]
        a[0] += float(r["Counter_Value"])
        a[1] += d[0]
        a[2] += 1
    tot_b = sum(a[0] for a in agg.values())
    tot_t = sum(a[1] for a in agg.values())
    per = sorted(({"kernel": k, "launches": a[2], "ms_per_step": round(a[1] / nsteps / 1e6, 3), "mfma_pipe_busy": round(a[0] / (1024 * a[1] * ghz), 4)}
                  for k, a in agg.items()), key=lambda x: -x["ms_per_step"])[:14]
    json.dump({"what": f"SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel time x {ghz} GHz) per kernel of bench.py --steps 2 --warmup 1 (one rocprofv3 --pmc pass, kernel-trace only); "
                       f"{nsteps} steps are traced; " + note,
               "per_kernel_mfma_busy": per, "whole_step_mfma_pipe_busy": round(tot_b / (1024 * tot_t * ghz), 4), "kernel_ms_per_step": round(tot_t / nsteps / 1e6, 3)}, open(out, "w"), indent=1)
    print(open(out).read()[:900])


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        busy(sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5], sys.argv[6])
