cd /root/repo
for v in "" ll_w6 ll_w7 ll_w8; do
  if [ -n "$v" ]; then export PTR_LIB=$PWD/ptranking_amd/libptranking_amd.$v.so; else unset PTR_LIB; fi
  echo "== ${v:-product}"; bash scratch/r5_kprof.sh 2>&1 | grep "lambdaloss_topk"
done
