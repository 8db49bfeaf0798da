#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r5
python scratch/r5_dbg_acts.py 2>&1 | grep -v amdgpu
timeout 1500 python -m pytest tests/test_x6_gpu.py tests/test_scorer_gpu.py tests/test_regime_gpu.py tests/test_ranker_gpu.py tests/test_dp_gpu.py tests/test_stack_gpu.py tests/test_bn_padded_gpu.py tests/test_ffnet_gpu.py tests/test_example_gpu.py tests/test_batching_gpu.py -q -m gpu 2>&1 | tail -6
