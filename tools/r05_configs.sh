#!/bin/bash
# The other geometries of SURVEY 8(d) on the round-5 build: bench.py --config X (no CPU baseline, no roofline replay), one JSON line each.
# usage (GPU box, repo root): bash tools/r05_configs.sh  ->  gpurun_out/r05_configs.log
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05_configs.log
: > $O
for spec in "A 8" "Bp 8" "C 8" "D 8" "E 2" "E 4" "B 1" "B 2" "B 4" "B 16"; do
  set -- $spec
  echo "== config $1, $2 clips per GPU" >> $O
  timeout -k 10 300 python3 bench.py --config $1 --batch $2 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep '^{' >> $O || echo "failed" >> $O
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r05_configs.log"):
    if l.startswith("=="): print(l.strip(), end="  ")
    elif l.startswith("{"):
        d = json.loads(l); print(d["value"], d["unit"], d["ms_per_step"], "ms", d.get("roofline"))
    else: print(l.strip())
PY
# secondary legs on the same box (GAN branch, the shipped 'sq' bottleneck, an FSQ autoencoder, the AR prior, the fused optimizer)
python3 bench.py --gan --sq --fsq-ae autoencoder_first_token_f256t512 --ar B --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' > gpurun_out/r05_secondary_legs.json || echo "secondary legs failed"
python3 bench.py --optimizer fused --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' > gpurun_out/r05_fused_optimizer.json || echo "optimizer leg failed"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r05_secondary_legs.json"))
for k in ("gan_step", "sq_step", "fsq_autoencoder_step", "ar_prior"):
    print(k, json.dumps(d.get(k))[:300])
d = json.load(open("gpurun_out/r05_fused_optimizer.json")); print("fused optimizer", d["value"], d["ms_per_step"])
PY
