#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r5
{ python scratch/exp_x6_ab.py
for v in x6_L0 x6_L1 x6_TM x6_TML1; do PTR_LIB=$PWD/ptranking_amd/libptranking_amd.$v.so python scratch/exp_x6_ab.py; done; } 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/c2_x6ab.log
cat gpurun_out/r5/c2_x6ab.log
timeout 600 python -m pytest tests/test_x6_gpu.py -x -q -m gpu 2>&1 | tail -3
python bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['value'], d.get('ms_per_step_at_1024'))"
