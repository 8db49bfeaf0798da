cd /root/repo
for m in 0 1 2 4; do echo "== PTR_PERSIST_MULT=$m"; PTR_PERSIST_MULT=$m bash scratch/r5_kprof.sh 2>&1 | grep "metrics_kernel\|lambdaloss_topk"; done
