#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6
for v in b6_trace2 b6_trace2_nolds; do echo "== $v"; TRACE2=1 PTR_LIB=$PWD/ptranking_amd/libptranking_amd.$v.so python scratch/exp_bwd_x6_trace.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r6/trace_bwd5.log
