# r5: stand-alone kernel times (rocprofv3 --kernel-trace --stats) + SQ_INSTS_VALU of profiles/prof_kernels.py run 65536
cd /root/repo
ROOT=$PWD; OUT=$ROOT/gpurun_out/r5k; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kstats --output-format csv -- python $ROOT/profiles/prof_kernels.py run 65536 > $OUT/kstats.log 2>&1
if [ "$1" = "valu" ]; then
rocprofv3 --pmc SQ_INSTS_VALU -d $OUT/kpmc --output-format csv -- python $ROOT/profiles/prof_kernels.py run 65536 > $OUT/kpmc.log 2>&1
fi
cd $ROOT
S=$(find $OUT/kstats -name '*kernel_stats.csv' | head -1)
V=$(find $OUT/kpmc -name '*counter_collection.csv' 2>/dev/null | head -1)
python - "$S" "$V" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
valu = {}
if len(sys.argv) > 2 and sys.argv[2]:
    sys.path.insert(0, "profiles")
    import pmc_traffic
    valu = pmc_traffic.per_kernel(sys.argv[2], "SQ_INSTS_VALU")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    if "ptr::" not in r["Name"]: continue
    nm = r["Name"].split("(")[0]
    v = next((x for k, x in valu.items() if nm.replace("void ", "") in k), None)
    print(f"{nm[:60]:60s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:8.1f} us" + (f"  VALU/wave {v / 65536:7.0f}" if v else ""))
P
