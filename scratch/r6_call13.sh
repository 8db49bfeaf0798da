#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6
for v in "" x6_xl1 x6_xl1q2 x6_xl1q3 x6_xl2 x6_xl2q2 ""; do
  if [ -z "$v" ]; then python scratch/r6_ab_fwd.py; else PTR_LIB=$PWD/ptranking_amd/libptranking_amd.$v.so python scratch/r6_ab_fwd.py; fi
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/ab_fwd13.log
