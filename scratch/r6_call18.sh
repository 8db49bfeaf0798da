#!/bin/bash
ROOT=/root/repo; cd $ROOT; mkdir -p gpurun_out/r6
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
PTR_REUSE_IMG=$mode rocprofv3 --kernel-trace --stats -d /tmp/st$mode --output-format csv -- python $ROOT/bench.py --batch 1024 --steps 60 --warmup 3 --no-cpu-baseline --sweep= --windows 1 --extras off > /tmp/st$mode.log 2>&1
f=$(find /tmp/st$mode -name "*kernel_stats.csv" | head -1); echo "== reuse=$mode"; python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    print(f"{r['Name'].split('(')[0][-45:]:46s} calls {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}")
PY
grep -o '"ms_per_step": [0-9.]*' /tmp/st$mode.log | head -1
done 2>&1 | tee $ROOT/gpurun_out/r6/kstats18.txt
