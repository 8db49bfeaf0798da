#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6
for v in "" b6_co b6_so4 b6_co_so4 "" b6_co; do
  if [ -z "$v" ]; then python scratch/r6_ab_bwd.py; else PTR_LIB=$PWD/ptranking_amd/libptranking_amd.$v.so python scratch/r6_ab_bwd.py; fi
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/ab_bwd3.log
for v in b6_trace_co b6_trace_co_so4; do echo "== $v"; PTR_LIB=$PWD/ptranking_amd/libptranking_amd.$v.so python scratch/exp_bwd_x6_trace.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r6/trace_bwd3.log
