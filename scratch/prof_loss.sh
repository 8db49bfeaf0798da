#!/bin/bash
# rocprofv3 kernel time of the LambdaRank loss kernel at L=128 / 256 (scratch/exp_loss2.py); PTR_LIB selects a variant build
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/pl; PTR_LIB=$lib rocprofv3 --kernel-trace --stats -d /tmp/pl --output-format csv -- python /root/repo/scratch/exp_loss2.py > /tmp/pl.log 2>&1
  echo "== $lib"; grep "pairs/s" /tmp/pl.log
  f=$(find /tmp/pl -name "*kernel_stats.csv" | head -1); python -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'ring' in r['Name'] or 'pairwise_bce' in r['Name']: print(r['Name'][:45], r['Calls'], r['AverageNs'], r['MinNs'])
" "$f"
done
