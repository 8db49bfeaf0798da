#!/bin/bash
# r6: reduce_partials_kernel with eight partials in flight: kernel time inside the step at 4096 and 1024 queries
cd /tmp && export TMPDIR=/tmp
for B in 4096 1024; do
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r6/rp_$B --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --batch $B --steps 30 --warmup 3 --no-cpu-baseline --sweep= --windows 1 --extras off > /dev/null 2>&1
python - $(find $GRAFT_REPO_ROOT/gpurun_out/r6/rp_$B -name '*kernel_stats.csv' | head -1) <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:5]:
    print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
cd $GRAFT_REPO_ROOT; python -m pytest tests/test_ranker_gpu.py tests/test_scorer_gpu.py -x -q -m gpu 2>&1 | tail -2
find gpurun_out/r6 -name '*.db' -delete; find gpurun_out/r6 -name '*kernel_trace.csv' -delete
