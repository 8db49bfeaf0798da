#!/bin/bash
ROOT=/root/repo; cd $ROOT; mkdir -p gpurun_out/r6
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/st1024 --output-format csv -- python $ROOT/bench.py --batch 1024 --steps 40 --warmup 3 --no-cpu-baseline --sweep= --windows 1 --extras off > /tmp/st1024.log 2>&1
f=$(find /tmp/st1024 -name "*kernel_stats.csv" | head -1); head -9 "$f" | cut -c1-60,150-260 | tee $ROOT/gpurun_out/r6/kstats17_B1024.txt
tail -2 /tmp/st1024.log | cut -c1-300
