#!/bin/bash
# BASELINE.json configs 1, 3, 4 (+ the default-pointsf step) as bench lines -> gpurun_out/r02/
mkdir -p gpurun_out/r02
python bench.py --loss RankNet --list-len 32 --batch 4096 --steps 50 --no-cpu-baseline --sweep= > gpurun_out/r02/r02_bench_c1_ranknet_L32.json 2>/dev/null
python bench.py --loss ListNet --list-len 256 --batch 4096 --steps 30 --no-cpu-baseline --sweep= > gpurun_out/r02/r02_bench_c3_listnet_L256.json 2>/dev/null
python bench.py --loss ListMLE --list-len 256 --batch 4096 --steps 30 --no-cpu-baseline --sweep= > gpurun_out/r02/r02_bench_c3_listmle_L256.json 2>/dev/null
python bench.py --loss ApproxNDCG --list-len 512 --features 700 --batch 1024 --steps 20 --nbatches 2 --no-cpu-baseline --sweep= > gpurun_out/r02/r02_bench_c4_approxndcg_L512_F700.json 2>/dev/null
python bench.py --loss LambdaRank --list-len 256 --batch 4096 --steps 30 --no-cpu-baseline --sweep= > gpurun_out/r02/r02_bench_northstar_lambdarank_L256.json 2>/dev/null
python bench.py --scorer pointsf_default --batch 1024 --steps 30 --warmup 5 --no-cpu-baseline --sweep= > gpurun_out/r02/r02_bench_default_pointsf_B1024.json 2>/dev/null
python bench.py --scorer listsf --loss LambdaLoss --list-len 256 --batch 1024 --steps 10 --warmup 2 > gpurun_out/r02/r02_bench_c5_listsf_lambdaloss_L256.json 2>/dev/null
for f in gpurun_out/r02/r02_bench_*.json; do python - "$f" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], round(j["value"]), "q/s", round(j["ms_per_step"],3), "ms/step")
PY
done
python scratch/exp_default_pointsf.py
