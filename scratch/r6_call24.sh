#!/bin/bash
# r6: default pointsf step (5 layers, BN + GELU) against the chunk count of the backward column sums (colsum2_kernel<1>)
mkdir -p gpurun_out/r6
for cap in 512 1024 2048 4096; do
  echo "PTR_BN_BWD_BLOCKS=$cap: $(PTR_BN_BWD_BLOCKS=$cap python bench.py --scorer pointsf_default --batch 1024 --steps 30 --warmup 5 --no-cpu-baseline --sweep= --extras off 2>/dev/null | python -c 'import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j["ms_per_step"],4), "ms/step")')"
done
cd /tmp && export TMPDIR=/tmp
for cap in 512 2048; do
PTR_BN_BWD_BLOCKS=$cap rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r6/dp_$cap --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --scorer pointsf_default --batch 1024 --steps 30 --warmup 5 --no-cpu-baseline --sweep= --windows 1 --extras off > /dev/null 2>&1
echo "cap $cap:"; python - $(find $GRAFT_REPO_ROOT/gpurun_out/r6/dp_$cap -name '*kernel_stats.csv' | head -1) <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f}%")
PY
done
find $GRAFT_REPO_ROOT/gpurun_out/r6 -name '*.db' -delete; find $GRAFT_REPO_ROOT/gpurun_out/r6 -name '*kernel_trace.csv' -delete
