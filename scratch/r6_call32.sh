#!/bin/bash
# r6: backward column sums of the batch norm: rows in flight (2 / 4) x chunk cap
for lib in "" bnr4; do for cap in 512 2048; do
  if [ -n "$lib" ]; then export PTR_LIB=ptranking_amd/libptranking_amd.$lib.so; else unset PTR_LIB; fi
  echo "rows=${lib:-2(product)} cap=$cap: $(PTR_BN_BWD_BLOCKS=$cap python bench.py --scorer pointsf_default --batch 1024 --steps 30 --warmup 5 --no-cpu-baseline --sweep= --extras off 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"],4), "ms/step")')"
done; done
unset PTR_LIB
python -m pytest tests/test_ffnet_gpu.py tests/test_stack_gpu.py tests/test_linear_gpu.py -x -q -m gpu 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r6/dp32 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --scorer pointsf_default --batch 1024 --steps 30 --warmup 5 --no-cpu-baseline --sweep= --windows 1 --extras off > /dev/null 2>&1
python - $(find $GRAFT_REPO_ROOT/gpurun_out/r6/dp32 -name '*kernel_stats.csv' | head -1) <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(f"  {r['Name'][:80]:80s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f}%")
PY
find $GRAFT_REPO_ROOT/gpurun_out/r6 -name '*.db' -delete; find $GRAFT_REPO_ROOT/gpurun_out/r6 -name '*kernel_trace.csv' -delete
