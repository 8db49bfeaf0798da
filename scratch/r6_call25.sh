#!/bin/bash
# r6: the generic linear layers at config 5's shapes: time per entry point and SQ counters per kernel
mkdir -p gpurun_out/r6
python scratch/exp_linear.py > gpurun_out/r6/lin_time.txt 2>&1
PTR_LIN_WIDE=0 python scratch/exp_linear.py > gpurun_out/r6/lin_time_wide0.txt 2>&1
scratch/prof_sq.sh gpurun_out/r6/lin_sq python $GRAFT_REPO_ROOT/scratch/exp_linear.py
cat gpurun_out/r6/lin_time.txt; echo; cat gpurun_out/r6/lin_time_wide0.txt; echo; grep -A16 "linear_fwd_kernel" gpurun_out/r6/lin_sq/sq_summary.txt | head -150
