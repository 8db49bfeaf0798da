#!/bin/bash
# round 5, call 1: tile-major activation stores (timing probes) + the bench line on this box
cd /root/repo
mkdir -p gpurun_out/r5
for v in "" x6_TM x6_TMH; do
  if [ -z "$v" ]; then python scratch/exp_x6_ab.py; else PTR_LIB=$PWD/ptranking_amd/libptranking_amd.$v.so python scratch/exp_x6_ab.py; fi
done > gpurun_out/r5/c1_x6ab.log 2>&1
python bench.py > gpurun_out/r5/c1_bench.json 2> gpurun_out/r5/c1_bench.err
cat gpurun_out/r5/c1_x6ab.log
python - <<'P'
import json
d=json.loads(open('gpurun_out/r5/c1_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('ms_per_step_at_1024'))
P
