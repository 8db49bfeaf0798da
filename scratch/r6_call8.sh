#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6
for t in 0 1; do echo "== TRAIN=$t"; TRAIN=$t PTR_LIB=$PWD/ptranking_amd/libptranking_amd.x6trace.so python scratch/exp_x6_trace.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r6/trace_fwd8.log
