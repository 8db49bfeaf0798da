#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8
python bench.py --no-cpu-baseline --extras off --sweep= > gpurun_out/r5/c5_bench_plain.json 2>/dev/null
python bench.py --no-cpu-baseline --extras off --sweep= --force-collectives > gpurun_out/r5/c5_bench_rccl1.json 2>gpurun_out/r5/c5_rccl1.err
python - <<'P'
import json
for f in ('plain','rccl1'):
    try:
        d=json.loads(open(f'gpurun_out/r5/c5_bench_{f}.json').read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], d['windows']['median_ms_per_step'], d['parallel'].get('allreduce_ms'))
    except Exception as e: print(f, 'failed', e)
P
tail -3 gpurun_out/r5/c5_rccl1.err
