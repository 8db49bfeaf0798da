#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_parity_gpu.py -q -m gpu -x -k "listwise or listnet or listmle" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_bench_contract.py -q -m gpu -x 2>&1 | tail -15
( time python bench.py > gpurun_out/r5/c4_bench.json 2> gpurun_out/r5/c4_bench.err ) 2>&1 | tail -4
tail -5 gpurun_out/r5/c4_bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r5/c4_bench.json').read().strip().splitlines()[-1])
print('step', d['ms_per_step'], d['value'])
print('padded', d.get('padded'))
mp=d.get('metric_path'); print('metric', {k:(v if not isinstance(v,dict) else {a:b for a,b in v.items() if a!='note' and a!='sample'}) for k,v in mp.items() if k!='what'} if mp else None)
print('configs', {k:(round(v['ms_per_step'],3), round(v['value'])) for k,v in d.get('configs',{}).items()})
P
