#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6
timeout 1200 python -m pytest tests/test_ranker_gpu.py tests/test_regime_gpu.py tests/test_dp_gpu.py tests/test_bench_contract.py tests/test_example_gpu.py -q -m gpu -x 2>&1 | tail -6
for B in 64 256 1024 4096; do python scratch/r6_small.py $B 200 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r6/small16.log
python bench.py --cpu-seconds 2 --extras off 2>gpurun_out/r6/bench16.err | tee gpurun_out/r6/bench16.json | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['value'], {k:(v['ms_per_step'] if 'ms_per_step' in v else v) for k,v in d['by_batch'].items()}); print({k:v.get('avg_launch_ms') for k,v in d['kernels'].items() if isinstance(v,dict)}, d['roofline']['avg_launch_ms'])"
tail -3 gpurun_out/r6/bench16.err
