#!/bin/bash
# r6: full GPU suite with the bf16x6 linear forward, then config 5 and the default pointsf step with it on / off
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for sw in 1 0; do
  echo "PTR_LIN_X6=$sw C5: $(PTR_LIN_X6=$sw python bench.py --scorer listsf --loss LambdaLoss --list-len 256 --batch 1024 --steps 10 --warmup 2 --windows 2 --no-cpu-baseline --sweep= 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"],3), "ms/step")')"
  echo "PTR_LIN_X6=$sw default pointsf B1024: $(PTR_LIN_X6=$sw python bench.py --scorer pointsf_default --batch 1024 --steps 30 --warmup 5 --no-cpu-baseline --sweep= --extras off 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"],4), "ms/step")')"
done
