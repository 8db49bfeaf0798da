#!/bin/bash
# r6: branch-free erf in the batch-norm / activation kernels: parity tests, default pointsf step and its kernels
python -m pytest tests/test_ffnet_gpu.py tests/test_stack_gpu.py tests/test_linear_gpu.py -x -q -m gpu 2>&1 | tail -3
echo "default pointsf B1024: $(python bench.py --scorer pointsf_default --batch 1024 --steps 30 --warmup 5 --no-cpu-baseline --sweep= --extras off 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"],4), "ms/step")')"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r6/dp31 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --scorer pointsf_default --batch 1024 --steps 30 --warmup 5 --no-cpu-baseline --sweep= --windows 1 --extras off > /dev/null 2>&1
python - $(find $GRAFT_REPO_ROOT/gpurun_out/r6/dp31 -name '*kernel_stats.csv' | head -1) <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print(f"  {r['Name'][:80]:80s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f}%")
PY
find $GRAFT_REPO_ROOT/gpurun_out/r6 -name '*.db' -delete; find $GRAFT_REPO_ROOT/gpurun_out/r6 -name '*kernel_trace.csv' -delete
