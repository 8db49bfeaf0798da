#!/bin/bash
cd /root/repo
python scratch/dbg_bwd_x6.py 2>&1 | grep -v amdgpu | tail -3
PTR_LIB=$PWD/ptranking_amd/libptranking_amd.b6noswz.so python scratch/dbg_bwd_x6.py 2>&1 | grep -v amdgpu | tail -1
python scratch/dbg_bwd_x6.py 2>&1 | grep -v amdgpu | tail -1
timeout 600 python -m pytest tests/test_x6_gpu.py -q -m gpu -x -k "bwd or backward" 2>&1 | tail -2
