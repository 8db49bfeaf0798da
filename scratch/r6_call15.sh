#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6
PTR_LIB=$PWD/ptranking_amd/libptranking_amd.b6_edge.so python scratch/r6_edge.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/edge15.log
python scratch/r6_ab_bwd.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_x6_gpu.py tests/test_scorer_gpu.py -q -m gpu -x -k "backward or bit_stable or grad" 2>&1 | tail -3
for B in 64 1024; do python scratch/r6_small.py $B 2>&1 | grep -v amdgpu.ids; done
