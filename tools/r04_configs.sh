#!/bin/bash
# The other geometries of SURVEY 8(d) on the round-4 build: bench.py --config X (no CPU baseline, no roofline replay), one JSON line each.
# usage (GPU box, repo root): bash tools/r04_configs.sh  ->  gpurun_out/r04_configs.log
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r04_configs.log
: > $O
for spec in "A 8" "Bp 8" "C 8" "D 8" "E 2" "E 4" "B 1" "B 2" "B 4" "B 16"; do
  set -- $spec
  echo "== config $1, $2 clips per GPU" >> $O
  timeout -k 10 300 python3 bench.py --config $1 --batch $2 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep '^{' >> $O || echo "failed" >> $O
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_configs.log"):
    if l.startswith("=="): print(l.strip(), end="  ")
    elif l.startswith("{"):
        d = json.loads(l); print(d["value"], d["unit"], d["ms_per_step"], "ms", d.get("roofline"))
    else: print(l.strip())
PY
