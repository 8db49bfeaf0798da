#!/bin/bash
# rocprofv3 kernel stats of the default-pointsf train step (scratch/exp_default_pointsf.py)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pd
rocprofv3 --kernel-trace --stats -d /tmp/pd --output-format csv -- python /root/repo/scratch/exp_default_pointsf.py > /tmp/pd.log 2>/dev/null
grep "ms/step" /tmp/pd.log
python /root/repo/scratch/kstats.py $(find /tmp/pd -name "*kernel_stats.csv") | head -${1:-24}
