#!/bin/bash
# r6 call 1: full GPU suite (strict-device default for gpu tests, ptr_train_step), backward variants A/B, phase trace, bench line
cd /root/repo; mkdir -p gpurun_out/r6
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r6/tests1.log
tail -5 gpurun_out/r6/tests1.log
for v in b6_r5 b6_w7only b6_tailonly "" dwpre b6_r5 ""; do
  if [ -z "$v" ]; then python scratch/r6_ab_bwd.py; else PTR_LIB=$PWD/ptranking_amd/libptranking_amd.$v.so python scratch/r6_ab_bwd.py; fi
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/ab_bwd1.log
PTR_LIB=$PWD/ptranking_amd/libptranking_amd.b6_trace.so python scratch/exp_bwd_x6_trace.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/trace_bwd1.log
python scratch/r6_ab_fwd.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/ab_fwd1.log
python bench.py --cpu-seconds 2 2>gpurun_out/r6/bench1.err | tee gpurun_out/r6/bench1.json | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['value'], {k:(v['ms_per_step'] if 'ms_per_step' in v else v) for k,v in d['by_batch'].items()})"
