#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6
for v in "" b6_st_noread b6_st_nomask b6_st_nowrite b6_st_nowrite_nomask b6_nostage ""; do
  if [ -z "$v" ]; then python scratch/r6_ab_bwd.py; else PTR_LIB=$PWD/ptranking_amd/libptranking_amd.$v.so python scratch/r6_ab_bwd.py; fi
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6/ab_bwd11.log
