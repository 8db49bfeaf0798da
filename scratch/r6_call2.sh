#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6
for v in b6_trace_ilv b6_trace_d3 b6_trace_ilv_d3; do echo "== $v"; PTR_LIB=$PWD/ptranking_amd/libptranking_amd.$v.so python scratch/exp_bwd_x6_trace.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r6/trace_bwd2.log
for B in 64 256 1024; do python scratch/r6_small.py $B 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r6/small2.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof64 --output-format csv -- python /root/repo/scratch/r6_small.py 64 200 > /dev/null 2>&1
f=$(find /tmp/prof64 -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-160 | tee /root/repo/gpurun_out/r6/kstats_B64.csv
