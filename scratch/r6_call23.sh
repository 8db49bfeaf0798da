#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6
for v in "" b6_em3 b6_em2 b6_em4 "" b6_em3; do
  if [ -z "$v" ]; then python scratch/r6_ab_bwd.py; else PTR_LIB=$PWD/ptranking_amd/libptranking_amd.$v.so python scratch/r6_ab_bwd.py; fi
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/ab_bwd23.log
TRACE2=1 PTR_LIB=$PWD/ptranking_amd/libptranking_amd.b6_em3_trace.so python scratch/exp_bwd_x6_trace.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/trace_bwd23.log
