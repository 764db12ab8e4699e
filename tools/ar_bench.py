"""LARP_AR (the prior over the tokenizer's indices) on one MI355X: training step (fwd + loss + bwd) and KV-cache generation.
The same leg as `bench.py --ar SIZE`, on its own.  usage: python tools/ar_bench.py [size=B] [batch=8] [seq=1024] [gen_batch=16] [mode=both|train|gen]"""
import json
import sys

sys.path.insert(0, ".")
import bench  # noqa: E402
import video_tokenizer_amd as vt  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "B"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
L = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
GB = int(sys.argv[4]) if len(sys.argv) > 4 else 16
mode = sys.argv[5] if len(sys.argv) > 5 else "both"
print(json.dumps(bench.ar_prior_step(vt, size, 10, 3, batch=B, seq=L, gen_batch=GB, mode=mode)))
