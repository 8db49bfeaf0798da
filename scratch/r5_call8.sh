cd /root/repo
for s in 20 50 100; do python bench.py --gpus 1 --steps $s --warmup 5 --windows 5 --no-cpu-baseline --sweep= --extras off 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($s, d['ms_per_step'], [round(x,4) for x in d['windows']['ms_per_step']])"; done
