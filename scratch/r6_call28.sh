#!/bin/bash
# r6: HBM traffic of the linear forward kernels at config 5's shapes (PMC, separate passes)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6/lin_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT/$c --output-format csv -- python $GRAFT_REPO_ROOT/scratch/exp_linear.py > $OUT/$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/r6/lin_pmc/{c}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c: agg[(r["Kernel_Name"][:60], r["Grid_Size"], r.get("LDS_Block_Size",""))].append(float(r["Counter_Value"]))
    print(c)
    for k, v in agg.items():
        if "linear_fwd" in k[0]: print("  ", k, "n", len(v), "mean", sum(v)/len(v))
PY
find $OUT -name '*.db' -delete; find $OUT -name '*counter_collection.csv' -delete
