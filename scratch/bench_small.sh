#!/bin/bash
# step time + forward/backward kernel time at the by_batch sizes (env passes through, e.g. PTR_FWD_WIDE=1)
for b in "$@"; do
  python bench.py --batch $b --steps 50 --warmup 10 --no-cpu-baseline --sweep= 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('B', $b, 'ms/step %.4f' % d['ms_per_step'], 'fwd %.1f us' % (k['scorer_forward']['avg_launch_ms']*1e3), 'bwd %.1f us' % (d['roofline']['avg_launch_ms']*1e3), 'loss %.1f us' % (k['lambdarank_loss_grad']['avg_launch_ms']*1e3))"
done
