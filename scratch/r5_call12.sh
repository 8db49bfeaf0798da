#!/bin/bash
cd /root/repo
python scratch/dbg_bwd_x6.py 2>&1 | grep -v amdgpu
PTR_LIB=$PWD/ptranking_amd/libptranking_amd.b6o1.so python scratch/dbg_bwd_x6.py 2>&1 | grep -v amdgpu | tail -2
timeout 600 python -m pytest tests/test_x6_gpu.py -q -m gpu -x -k "bwd or backward" 2>&1 | tail -3
