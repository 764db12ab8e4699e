#!/bin/bash
# One and two clips per GPU: the training step (fwd + loss + bwd) eager and as one hipGraph replay, with K split in the N = D GEMMs
# (the default) and without (VT_GEMM_SPLITK=0).  One JSON line per run.
out=gpurun_out/r3_small_batch.jsonl
: > $out
for b in 1 2 4; do
  for sk in 1 0; do
    for g in "" "--graph"; do
      VT_GEMM_SPLITK=$sk timeout -k 10 200 python bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-roofline $g 2>/dev/null | tail -1 | sed "s/^{/{\"split_k\": $sk, \"graph\": \"$g\", /" >> $out || exit 1
    done
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r3_small_batch.jsonl"):
    d = json.loads(l)
    print(d["config"].get("clips_per_gpu", d["config"]), "split_k", d["split_k"], "graph", bool(d["graph"]), d["ms_per_step"], "ms", d["value"], d["unit"])
PY
