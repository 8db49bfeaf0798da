#!/bin/bash
# SQ instruction counters of the LambdaRank loss kernel (scratch/exp_loss2.py) for each library given
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/pc; PTR_LIB=$lib ITERS=5 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES -d /tmp/pc --output-format csv -- python /root/repo/scratch/exp_loss2.py > /tmp/pc.log 2>&1
  echo "== $lib"
  python - $(find /tmp/pc -name "*counter_collection.csv" | head -1) <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if 'ring' in r['Kernel_Name']: acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(k, {c: round(sum(x)/len(x)) for c,x in sorted(v.items())})
PY
done
