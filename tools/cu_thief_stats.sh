#!/bin/bash
# per-kernel average durations of the step with 0 and with 8 CUs held (tools/cu_thief_steps.py under rocprofv3 --kernel-trace --stats)
# VT_THIEF_DP=1: with the data-parallel schedule (engine.set_wgrad_tail(3): block-by-block tail, multi-round GEMMs one tile per workgroup)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
export VT_THIEF_DP=${VT_THIEF_DP:-0}
for n in 0 8; do
  rm -rf /tmp/thief_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/thief_$n -- python3 $R/tools/cu_thief_steps.py $n 6 > $R/gpurun_out/cu_thief_stats_$n.log 2>&1 || echo "trace $n failed"
  cp $(find /tmp/thief_$n -name "*kernel_stats.csv" | head -1) $R/gpurun_out/cu_thief_kernel_stats_$n.csv
done
python3 - <<PY
import csv
def load(p):
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(p))}
a, b = load("$R/gpurun_out/cu_thief_kernel_stats_0.csv"), load("$R/gpurun_out/cu_thief_kernel_stats_8.csv")
print(f"{'kernel':70s} calls   avg us free   avg us 8 held   ratio   extra ms (6 steps)")
for k, (n, us, tot) in sorted(a.items(), key=lambda kv: -kv[1][2])[:14]:
    if k in b:
        print(f"{k[:70]:70s} {n:5d} {us:12.1f} {b[k][1]:14.1f} {b[k][1] / us:7.2f} {b[k][2] - tot:10.2f}")
PY
