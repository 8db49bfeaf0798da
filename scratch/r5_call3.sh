#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "listwise or listnet or listmle or knife" 2>&1 | tail -8
timeout 300 python -m pytest tests/test_siblings_gpu.py tests/test_ranker_gpu.py -q -m gpu -x 2>&1 | tail -3
python scratch/r5_listwise.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/c3_listwise.log
PTR_LISTNET_VEC=0 PTR_LISTMLE_VEC=0 python scratch/r5_listwise.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r5/c3_listwise.log
