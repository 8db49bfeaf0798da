#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_ranknet_pack_gpu.py -q -m gpu -x 2>&1 | tail -3
python scratch/r6_ring.py 2>&1 | grep -v "amdgpu.ids\|Warn\|Consider\|out\[ring" | tee gpurun_out/r6/ring7.log
