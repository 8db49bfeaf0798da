cd /root/repo
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_regime_gpu.py -x -q -m gpu -k "lambdaloss or metrics or sort or golden or knife or listwise or evaluator" 2>&1 | tail -3
bash scratch/r5_kprof.sh valu 2>&1 | tail -16
