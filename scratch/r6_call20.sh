#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6
timeout 1200 python -m pytest tests/test_x6_gpu.py tests/test_scorer_gpu.py tests/test_regime_gpu.py -q -m gpu -x 2>&1 | tail -3
for v in x6_kt0 "" x6_kt0 ""; do
  if [ -z "$v" ]; then python scratch/r6_ab_fwd.py; else PTR_LIB=$PWD/ptranking_amd/libptranking_amd.$v.so python scratch/r6_ab_fwd.py; fi
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/ab_fwd20.log
