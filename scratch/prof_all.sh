#!/bin/bash
# Round-2 measurement set on one MI355X: bench line, rocprofv3 kernel stats, PMC HBM traffic (separate passes), SQ counters.
# Outputs under gpurun_out/r02/ ; the summaries that are kept are copied into profiles/ by hand.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r02; mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.err
BENCH="python $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --sweep="
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --sweep= > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT/pmc_$c --output-format csv -- $BENCH > $OUT/pmc_$c.log 2>&1
done
cd $ROOT
python profiles/pmc_traffic.py $(find $OUT/pmc_FETCH_SIZE -name '*counter_collection.csv') $(find $OUT/pmc_WRITE_SIZE -name '*counter_collection.csv') 4096 128 136 $OUT/r02_pmc_traffic.json
scratch/prof_sq.sh gpurun_out/r02/sq $BENCH
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/r02_kernel_stats.csv
find $OUT -name '*.db' -delete; find $OUT -name '*kernel_trace.csv' -size +1M -delete; find $OUT -name '*counter_collection.csv' -size +1M -delete
head -c 1500 $OUT/bench.json; echo; head -12 $OUT/r02_kernel_stats.csv | cut -c1-150
