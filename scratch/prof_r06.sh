#!/bin/bash
# Round-6 measurement set on one MI355X (run from the repo root through gpurun): bench lines, rocprofv3 kernel stats, PMC HBM traffic
# (FETCH_SIZE / WRITE_SIZE in separate passes, no trace domains beside --pmc), SQ counters, the BASELINE configs, the stand-alone kernel
# profile.  Everything lands under gpurun_out/r06/; the summaries that are kept are copied into profiles/ by hand.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06; mkdir -p $OUT
python bench.py --batch 1024 --steps 100 --no-cpu-baseline --sweep= --extras off > $OUT/r06_bench_B1024.json 2>/dev/null
BENCH="python $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --sweep= --windows 1 --extras off"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --sweep= --windows 1 --extras off > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/stats1024 --output-format csv -- python $ROOT/bench.py --batch 1024 --steps 20 --warmup 3 --no-cpu-baseline --sweep= --windows 1 --extras off > $OUT/stats1024.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT/pmc_$c --output-format csv -- $BENCH > $OUT/pmc_$c.log 2>&1
done
cd $ROOT
python profiles/pmc_traffic.py $(find $OUT/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find $OUT/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1) 4096 128 136 $OUT/r06_pmc_traffic.json
# the headline line AFTER the PMC pass: bench.py trusts profiles/r06_pmc_traffic.json only when its kernel-source hash matches the built library
cp $OUT/r06_pmc_traffic.json profiles/r06_pmc_traffic.json
python bench.py > $OUT/r06_bench_B4096.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.err
scratch/prof_sq.sh gpurun_out/r06/sq $BENCH
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/r06_bench_B4096_kernel_stats.csv
cp $(find $OUT/stats1024 -name '*kernel_stats.csv' | head -1) $OUT/r06_bench_B1024_kernel_stats.csv
cp $OUT/sq/sq_summary.txt $OUT/r06_sq_c2.txt
# BASELINE configs
python bench.py --loss RankNet --list-len 32 --batch 4096 --steps 50 --no-cpu-baseline --sweep= > $OUT/r06_bench_c1_ranknet_L32.json 2>/dev/null
python bench.py --loss ListNet --list-len 256 --batch 4096 --steps 30 --no-cpu-baseline --sweep= > $OUT/r06_bench_c3_listnet_L256.json 2>/dev/null
python bench.py --loss ListMLE --list-len 256 --batch 4096 --steps 30 --no-cpu-baseline --sweep= > $OUT/r06_bench_c3_listmle_L256.json 2>/dev/null
python bench.py --loss ApproxNDCG --list-len 512 --features 700 --batch 1024 --steps 20 --nbatches 2 --no-cpu-baseline --sweep= > $OUT/r06_bench_c4_approxndcg_L512_F700.json 2>/dev/null
python bench.py --loss LambdaRank --list-len 256 --batch 4096 --steps 30 --no-cpu-baseline --sweep= > $OUT/r06_bench_northstar_lambdarank_L256.json 2>/dev/null
python bench.py --scorer pointsf_default --batch 1024 --steps 30 --warmup 5 --no-cpu-baseline --sweep= > $OUT/r06_bench_default_pointsf_B1024.json 2>/dev/null
python bench.py --scorer listsf --loss LambdaLoss --list-len 256 --batch 1024 --steps 10 --warmup 2 --windows 2 > $OUT/r06_bench_c5_listsf_lambdaloss_L256.json 2>/dev/null
# the launcher path: --gpus 1 with the RCCL process group of one (collectives executed), and the A/B of the bf16x6 forward
python bench.py --force-collectives --no-cpu-baseline --sweep= --extras off > $OUT/r06_bench_B4096_rccl_group_of_one.json 2>/dev/null
PTR_MLP_X6=0 python bench.py --no-cpu-baseline --sweep= --extras off > $OUT/r06_bench_B4096_fp32_mfma_forward.json 2>/dev/null
# kernel stats of config 5 and of the default-pointsf step
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/c5stats --output-format csv -- python $ROOT/bench.py --scorer listsf --loss LambdaLoss --list-len 256 --batch 1024 --steps 6 --warmup 2 --windows 1 --no-cpu-baseline > $OUT/c5stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/dpstats --output-format csv -- python $ROOT/bench.py --scorer pointsf_default --batch 1024 --steps 30 --warmup 5 --no-cpu-baseline --sweep= --windows 1 > $OUT/dpstats.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/c4stats --output-format csv -- python $ROOT/bench.py --loss ApproxNDCG --list-len 512 --features 700 --batch 1024 --steps 10 --warmup 2 --nbatches 2 --windows 1 --no-cpu-baseline --sweep= > $OUT/c4stats.log 2>&1
cd $ROOT
cp $(find $OUT/c4stats -name '*kernel_stats.csv' | head -1) $OUT/r06_c4_step_kernel_stats.csv
cp $(find $OUT/c5stats -name '*kernel_stats.csv' | head -1) $OUT/r06_c5_listsf_step_kernel_stats.csv
cp $(find $OUT/dpstats -name '*kernel_stats.csv' | head -1) $OUT/r06_default_pointsf_step_kernel_stats.csv
# stand-alone kernels at 65 536 queries
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/kstats --output-format csv -- python $ROOT/profiles/prof_kernels.py run 65536 > $OUT/kstats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  rocprofv3 --pmc $c -d $OUT/kpmc_$c --output-format csv -- python $ROOT/profiles/prof_kernels.py run 65536 > $OUT/kpmc_$c.log 2>&1
done
cd $ROOT
cp $(find $OUT/kstats -name '*kernel_stats.csv' | head -1) $OUT/r06_kernels_B65536_kernel_stats.csv
python profiles/prof_kernels.py summarise $OUT/r06_kernels_B65536_kernel_stats.csv $OUT/r06_kernels_B65536.json 65536 $(find $OUT/kpmc_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find $OUT/kpmc_WRITE_SIZE -name '*counter_collection.csv' | head -1) $(find $OUT/kpmc_SQ_INSTS_VALU -name '*counter_collection.csv' | head -1)
find $OUT -name '*.db' -delete; find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -size +1M -delete; find $OUT -name '*_agent_info.csv' -delete
for f in $OUT/r06_bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(j["value"]), "q/s", round(j["ms_per_step"],3), "ms/step", "roofline", round(j["roofline"].get("frac",0),3))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
head -8 $OUT/r06_bench_B4096_kernel_stats.csv | cut -c1-140
cat $OUT/r06_sq_c2.txt | head -30
