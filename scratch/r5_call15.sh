#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests/test_x6_gpu.py tests/test_scorer_gpu.py tests/test_ranker_gpu.py tests/test_dp_gpu.py tests/test_regime_gpu.py -q -m gpu -x 2>&1 | tail -4
python bench.py --no-cpu-baseline --extras off > gpurun_out/r5/c15_bench.json 2>/dev/null
PTR_BWD_X6=0 python bench.py --no-cpu-baseline --extras off --sweep= > gpurun_out/r5/c15_bench_fp32bwd.json 2>/dev/null
python - <<'P'
import json
for f in ('c15_bench','c15_bench_fp32bwd'):
    d=json.loads(open(f'gpurun_out/r5/{f}.json').read().strip().splitlines()[-1])
    print(f, 'step', d['ms_per_step'], d['windows']['median_ms_per_step'], 'fwd', d['kernels']['scorer_forward']['avg_launch_ms'], 'bwd', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline'].get('frac_of_fp32_mfma_peak'), 'b1024', d.get('ms_per_step_at_1024'), {k:round(v['ms_per_step'],4) for k,v in d['by_batch'].items()})
P
