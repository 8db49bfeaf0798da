#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r06
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --sweep= --windows 1 --extras off 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('smoke bench', d['ms_per_step'], d['step_entry'][:40])" || exit 1
python bench.py > gpurun_out/r06/r06_bench_B4096.json 2> gpurun_out/r06/bench.err; tail -c 300 gpurun_out/r06/bench.err; python -c "import json; d=json.loads(open(\"gpurun_out/r06/r06_bench_B4096.json\").read().strip().splitlines()[-1]); print(d[\"ms_per_step\"], d[\"value\"], d[\"value_at_1024\"], d[\"roofline\"][\"frac\"], d[\"roofline\"][\"traffic\"], {k:v[\"ms_per_step\"] for k,v in d[\"by_batch\"].items()})"
tail -60 gpurun_out/r06/prof.log
