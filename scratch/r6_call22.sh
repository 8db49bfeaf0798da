#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r6
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8
