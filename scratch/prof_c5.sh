#!/bin/bash
# rocprofv3 kernel stats of the C5 step (listsf + LambdaLoss, L=256, 1024 queries) -> gpurun_out/r02/c5_kernel_stats.csv
ROOT=$(pwd); mkdir -p $ROOT/gpurun_out/r02
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p5
rocprofv3 --kernel-trace --stats -d /tmp/p5 --output-format csv -- python $ROOT/bench.py --scorer listsf --loss LambdaLoss --list-len 256 --batch 1024 --steps 8 --warmup 2 --no-cpu-baseline --sweep= > /tmp/p5.log 2>/dev/null
tail -1 /tmp/p5.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'])"
cp $(find /tmp/p5 -name "*kernel_stats.csv") $ROOT/gpurun_out/r02/c5_kernel_stats.csv
python $ROOT/scratch/kstats.py $ROOT/gpurun_out/r02/c5_kernel_stats.csv | head -${1:-30}
