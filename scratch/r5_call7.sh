cd /root/repo
ROOT=$PWD; OUT=$ROOT/gpurun_out/r5p; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU"; do
  d=$OUT/$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c -d $d --output-format csv -- python $ROOT/profiles/prof_kernels.py run 65536 > $d.log 2>&1 || tail -3 $d.log
done
cd $ROOT
python - <<'P'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r5p/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if any(x in k for x in ("lambdaloss_topk", "metrics_kernel", "sort_desc", "shuffle_ties")):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:26s} {sum(v) / len(v):14.0f}  (n={len(v)})")
P
