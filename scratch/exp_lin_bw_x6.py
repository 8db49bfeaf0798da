"""A/B of ptr_linear_backward_weight: fp32-MFMA kernel (PTR_LIN_BW_X6=0) vs the bf16x6 narrow-side kernel (=2) on the listsf / default-pointsf shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptranking_amd.linear import _bwd_weight
for R, K, N in ((262144, 136, 408), (262144, 136, 136), (262144, 136, 128), (262144, 128, 256), (131072, 100, 100), (262144, 512, 136), (262144, 256, 512)):
    x = torch.randn(R, K, device="cuda"); dy = torch.randn(R, N, device="cuda")
    dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
    for mode in ("0", "2"):
        os.environ["PTR_LIN_BW_X6"] = mode
        for _ in range(3): _bwd_weight(x, K, dy, True, dw_out=dw, db_out=db)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): _bwd_weight(x, K, dy, True, dw_out=dw, db_out=db)
        e1.record(); torch.cuda.synchronize()
        print(f"R={R} K={K} N={N} PTR_LIN_BW_X6={mode}: {e0.elapsed_time(e1) * 100:8.1f} us", flush=True)
