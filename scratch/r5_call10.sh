#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "lambdaloss" 2>&1 | tail -12
timeout 600 python -m pytest tests/test_ranker_gpu.py tests/test_listsf_gpu.py -q -m gpu -x 2>&1 | tail -3
python scratch/r5_small.py 2>&1 | grep -v amdgpu | grep -i "lambdaloss\|shuffle"
PTR_LAMBDALOSS_TOPK=0 python scratch/r5_small.py 2>&1 | grep -v amdgpu | grep -i "lambdaloss"
