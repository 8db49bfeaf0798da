#!/bin/bash
cd /root/repo
python scratch/dbg_bwd_x6.py 2>&1 | grep -v amdgpu | tail -1
for v in b6prio; do echo $v; PTR_LIB=$PWD/ptranking_amd/libptranking_amd.$v.so python scratch/dbg_bwd_x6.py 2>&1 | grep -v amdgpu | tail -1; done
