#!/bin/bash
# stock PyTorch-ROCm step and per-op yardsticks next to bench.py on the same box -> gpurun_out/torch_yardstick.log
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/torch_yardstick.log
: > $O
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep '^{' >> $O
timeout -k 10 300 python3 tools/torch_yardstick.py --batch 8 --steps 10 >> $O 2>gpurun_out/torch_yardstick.err || echo '{"error": "eager yardstick failed"}' >> $O
timeout -k 10 200 python3 tools/torch_yardstick.py --batch 8 --steps 10 --deterministic >> $O 2>>gpurun_out/torch_yardstick.err || echo '{"error": "deterministic yardstick failed"}' >> $O
timeout -k 10 200 python3 tools/torch_yardstick.py --batch 1 --steps 20 >> $O 2>>gpurun_out/torch_yardstick.err || echo '{"error": "1-clip yardstick failed"}' >> $O
timeout -k 10 300 python3 tools/torch_yardstick.py --ops >> $O 2>>gpurun_out/torch_yardstick.err || echo '{"error": "ops yardstick failed"}' >> $O
timeout -k 10 600 python3 tools/torch_yardstick.py --batch 8 --steps 10 --compile >> $O 2>>gpurun_out/torch_yardstick.err || echo '{"error": "torch.compile yardstick failed or timed out"}' >> $O
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep '^{' >> $O
cut -c1-400 $O
tail -5 gpurun_out/torch_yardstick.err
