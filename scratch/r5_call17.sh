#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests/test_x6_gpu.py tests/test_scorer_gpu.py tests/test_regime_gpu.py tests/test_ranker_gpu.py -q -m gpu -x 2>&1 | tail -4
python bench.py --loss ApproxNDCG --list-len 512 --features 700 --batch 1024 --steps 20 --nbatches 2 --no-cpu-baseline --sweep= --extras off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 x6 tail', d['ms_per_step'], d['value'])"
PTR_BWD_X6=0 python bench.py --loss ApproxNDCG --list-len 512 --features 700 --batch 1024 --steps 20 --nbatches 2 --no-cpu-baseline --sweep= --extras off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 fp32 tail', d['ms_per_step'], d['value'])"
python bench.py --features 48 --steps 50 --no-cpu-baseline --sweep= --extras off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('F=48 x6 tail', d['ms_per_step'], d['value'])"
PTR_BWD_X6=0 python bench.py --features 48 --steps 50 --no-cpu-baseline --sweep= --extras off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('F=48 fp32 tail', d['ms_per_step'], d['value'])"
