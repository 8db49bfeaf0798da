#!/bin/bash
# r6: linear_fwd_x6_kernel: 8 waves x 32-document tiles against 16 waves x 16-document tiles
for v in "" lxform1; do
  echo "== ${v:-product}"; if [ -n "$v" ]; then export PTR_LIB=ptranking_amd/libptranking_amd.$v.so; fi
  python scratch/exp_linear.py 2>&1 | grep -E "K= 136 N= 136|K= 128 N= 256|K= 136 N= 408|K= 136 N= 128"
done
