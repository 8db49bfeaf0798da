#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6
timeout 600 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "golden" 2>&1 | tail -2
for v in "" b6_pre "" b6_pre; do
  if [ -z "$v" ]; then python scratch/r6_ab_bwd.py; else PTR_LIB=$PWD/ptranking_amd/libptranking_amd.$v.so python scratch/r6_ab_bwd.py; fi
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/ab_bwd21.log
TRACE2=1 PTR_LIB=$PWD/ptranking_amd/libptranking_amd.b6_pre_trace.so python scratch/exp_bwd_x6_trace.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/trace_bwd21.log
