#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests/test_x6_gpu.py tests/test_scorer_gpu.py tests/test_regime_gpu.py tests/test_ranker_gpu.py tests/test_dp_gpu.py tests/test_stack_gpu.py -q -m gpu -x 2>&1 | tail -8
python scratch/exp_x6_ab.py 2>&1 | grep -v amdgpu.ids
python bench.py --no-cpu-baseline --extras off > gpurun_out/r5/c6_bench.json 2>/dev/null
python - <<'P'
import json
d=json.loads(open('gpurun_out/r5/c6_bench.json').read().strip().splitlines()[-1])
print('step', d['ms_per_step'], d['windows']['median_ms_per_step'], 'fwd', d['kernels']['scorer_forward']['avg_launch_ms'], 'bwd', d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'b1024', d.get('ms_per_step_at_1024'))
P
