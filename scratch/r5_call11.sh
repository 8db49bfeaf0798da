#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests/test_scorer_gpu.py tests/test_ranker_gpu.py tests/test_dp_gpu.py tests/test_x6_gpu.py -q -m gpu -x 2>&1 | tail -3
python bench.py --no-cpu-baseline --extras off > gpurun_out/r5/c11_bench.json 2>/dev/null
PTR_REDUCE4=0 python bench.py --no-cpu-baseline --extras off --sweep= > gpurun_out/r5/c11_bench_old.json 2>/dev/null
python - <<'P'
import json
for f in ('c11_bench','c11_bench_old'):
    d=json.loads(open(f'gpurun_out/r5/{f}.json').read().strip().splitlines()[-1])
    print(f, 'step', d['ms_per_step'], d['windows']['median_ms_per_step'], 'fwd', d['kernels']['scorer_forward']['avg_launch_ms'], 'bwd', d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'b1024', d.get('ms_per_step_at_1024'), 'b64', d['by_batch'].get('64',{}).get('ms_per_step'))
P
