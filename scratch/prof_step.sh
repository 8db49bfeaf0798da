#!/bin/bash
# rocprofv3 kernel stats of the bench step: scratch/prof_step.sh BATCH [pattern ...]   (env passes through)
b=$1; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ps
rocprofv3 --kernel-trace --stats -d /tmp/ps --output-format csv -- python /root/repo/bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline --sweep= > /tmp/ps.log 2>/dev/null
python /root/repo/scratch/kstats.py $(find /tmp/ps -name "*kernel_stats.csv") "$@"
