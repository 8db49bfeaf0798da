#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_sort_gpu.py tests/test_siblings_gpu.py -q -m gpu -x 2>&1 | tail -3
python scratch/r5_small.py 2>&1 | grep -v amdgpu
