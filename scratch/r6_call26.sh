#!/bin/bash
# r6: bf16x6 generic linear forward / backward-input: parity tests, then time at config 5's shapes (whole-tile form / general form / fp32)
mkdir -p gpurun_out/r6
python -m pytest tests/test_linear_gpu.py tests/test_ffnet_gpu.py tests/test_stack_gpu.py tests/test_listsf_gpu.py -x -q -m gpu 2>&1 | tail -4
echo "== x6 (whole-tile form where it serves)"; python scratch/exp_linear.py 2>&1 | grep -v amdgpu.ids
echo "== x6 general form only (PTR_LIN_X6=2)"; PTR_LIN_X6=2 python scratch/exp_linear.py 2>&1 | grep -v amdgpu.ids
echo "== fp32 MFMA (PTR_LIN_X6=0)"; PTR_LIN_X6=0 python scratch/exp_linear.py 2>&1 | grep -v amdgpu.ids
