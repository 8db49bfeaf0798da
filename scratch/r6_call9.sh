#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_x6_gpu.py tests/test_regime_gpu.py -q -m gpu -s -k "error_not_above or backward_matches_float64 or headline_batch" 2>&1 | grep -E "MEASURED|passed|failed|Error|assert" | tee gpurun_out/r6/measured9.log | tail -80
echo "=== golden families at EL_RTOL = 1e-5"
PTR_GOLDEN_EL_RTOL=1e-5 timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_siblings_gpu.py tests/test_ffnet_gpu.py tests/test_listsf_gpu.py tests/test_bn_padded_gpu.py tests/test_approx_ring_gpu.py tests/test_ranknet_pack_gpu.py tests/test_oracle_golden.py -q -m gpu 2>&1 | grep -E "^FAILED|passed|failed|element-wise" | tee gpurun_out/r6/golden_1e5.log | tail -60
