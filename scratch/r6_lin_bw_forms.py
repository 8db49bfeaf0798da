"""r6: ptr_linear_backward_weight on the listsf / default-pointsf shapes: fp32-MFMA kernel (PTR_LIN_BW_X6=0), bf16x6 with the 8-wave / 24-tile form (PTR_LIN_BW_FORM=24) and
with the 16-wave / 16-tile form (default) for products of 9+ wide tiles; products with <= 8 wide tiles run the one-tile-per-wave form either way."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptranking_amd.linear import _bwd_weight
for R, K, N in ((262144, 136, 408), (262144, 136, 136), (262144, 136, 128), (262144, 128, 256), (131072, 100, 100), (262144, 512, 136), (262144, 256, 512)):
    x = torch.randn(R, K, device="cuda"); dy = torch.randn(R, N, device="cuda")
    dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
    ref = None; row = []
    for mode, form in (("0", "16"), ("2", "24"), ("2", "16")):
        os.environ["PTR_LIN_BW_X6"] = mode; os.environ["PTR_LIN_BW_FORM"] = form
        for _ in range(3): _bwd_weight(x, K, dy, True, dw_out=dw, db_out=db)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): _bwd_weight(x, K, dy, True, dw_out=dw, db_out=db)
        e1.record(); torch.cuda.synchronize()
        if ref is None: ref = (dw.clone(), db.clone())
        d = float((dw - ref[0]).abs().max() / ref[0].abs().max()); d2 = float((db - ref[1]).abs().max() / ref[1].abs().max())
        row.append(f"{'fp32' if mode == '0' else 'x6/' + form}: {e0.elapsed_time(e1) * 100:7.1f} us (vs fp32 {d:.1e} / {d2:.1e})")
    print(f"R={R} K={K} N={N}: " + "  ".join(row), flush=True)
