#!/bin/bash
# config 5 and the default pointsf step with both tile forms of the linear forward / backward-input kernel
for w in 1 0; do
  PTR_LIN_WIDE=$w python bench.py --scorer listsf --loss LambdaLoss --list-len 256 --batch 1024 --steps 6 --warmup 2 --windows 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PTR_LIN_WIDE=$w C5', j['windows']['ms_per_step'])"
  PTR_LIN_WIDE=$w python bench.py --scorer pointsf_default --batch 1024 --steps 30 --warmup 5 --no-cpu-baseline --sweep= --windows 2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PTR_LIN_WIDE=$w default pointsf', j['windows']['ms_per_step'])"
done
