#!/bin/bash
# Long runs on the final build (one box): ms per step, clips/s; the loss must stay finite (bench.py asserts it).  -> gpurun_out/r04_soak.log
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r04_soak.log
echo "# Long runs on the final round-4 build (one box, bench.py --no-cpu-baseline --no-roofline): ms per step, clips/s.  Losses finite (asserted by bench.py)." > $O
for args in "--batch 8 --steps 300 --warmup 5 --optimizer fused" "--batch 8 --steps 200 --warmup 5 --force-dist" "--batch 1 --steps 400 --warmup 5 --optimizer fused" "--batch 1 --steps 400 --warmup 5 --graph" "--batch 2 --steps 300 --warmup 5 --optimizer fused" "--batch 8 --steps 100 --warmup 5 --gan"; do
  timeout -k 10 400 python3 bench.py $args --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
extra = ''
if 'gan_step' in d: extra = '   gan_step: ' + json.dumps(d['gan_step'])[:160]
print('%-55s %8.3f ms %8.1f clips/s%s' % ('$args', d['ms_per_step'], d['value'], extra))" >> $O || echo "$args: FAILED" >> $O
done
cat $O
