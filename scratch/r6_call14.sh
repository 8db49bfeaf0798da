#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6
PTR_LIB=$PWD/ptranking_amd/libptranking_amd.b6_edge.so python scratch/r6_edge.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/edge14.log
