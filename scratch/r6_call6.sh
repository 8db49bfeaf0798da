#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "pairwise or ring or knife or golden" 2>&1 | tail -4
python scratch/r6_ring.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/ring6.log
