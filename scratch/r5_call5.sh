cd /root/repo
mkdir -p gpurun_out/r5
timeout 300 python scratch/r5_sortk.py 2>&1 | tee gpurun_out/r5/sortk.log | tail -6
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_regime_gpu.py -x -q -m gpu -k "lambdaloss or metrics or sort or golden or knife or listwise or evaluator" 2>&1 | tail -3
