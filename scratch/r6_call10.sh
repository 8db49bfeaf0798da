#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6
timeout 2400 python -m pytest tests -q -m gpu -s 2>&1 | grep -E "MEASURED|passed|failed|^FAILED|Error" > gpurun_out/r6/measured10.log
grep -E "passed|failed|^FAILED" gpurun_out/r6/measured10.log | tail -20
grep -E "x6-wide-dw" gpurun_out/r6/measured10.log | awk '{print $0}' | sort -t' ' -k12 | tail -8
grep -E "bench-scale" gpurun_out/r6/measured10.log
