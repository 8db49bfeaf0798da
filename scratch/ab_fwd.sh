#!/bin/bash
# A/B of two library builds on one box: scorer forward / backward launch averages inside the bench step.  usage: ab_fwd.sh libA.so libB.so
for rep in 1 2 3; do for lib in "$@"; do
  PTR_LIB=$lib python bench.py --steps 100 --no-cpu-baseline --sweep= --windows 2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib'.split('/')[-1], 'step', round(j['ms_per_step'],4), 'fwd', round(j['kernels']['scorer_forward']['avg_launch_ms']*1e3,1), 'us  bwd entry', round(j['roofline']['avg_launch_ms']*1e3,1), 'us')
"; done; done
