"""Layer 1 of the pointsf scorer from six bf16 matrix instructions per 32-deep slice (scratch/bf16x6/l1.hip -> l1.hsaco): accuracy against float64 and
time next to the product's fp32-MFMA linear kernel on the same shape (R = 524 288 rows, 136 -> 100 features, ReLU)."""
import ctypes as C, os, sys, torch, numpy as np
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
hip = C.CDLL("libamdhip64.so")
mod = C.c_void_p()
assert hip.hipModuleLoad(C.byref(mod), os.path.join(here, "l1.hsaco").encode()) == 0
def fn(name):
    f = C.c_void_p(); assert hip.hipModuleGetFunction(C.byref(f), mod, name.encode()) == 0; return f
def ptr(t): return C.cast(C.pointer(C.c_void_p(t.data_ptr())), C.c_void_p)
def ival(v): return C.cast(C.pointer(C.c_int(v)), C.c_void_p)
KP, LDW = 160, 168
LDS = 3 * 112 * LDW * 2
torch.manual_seed(0)
F, NH = 136, 100
W = torch.randn(NH, F) / F ** 0.5
b = torch.randn(NH) * 0.1
Wpad = torch.zeros(112, KP); Wpad[:NH, :F] = W
def hi16(x): return (x.view(torch.int32) >> 16).to(torch.int16)
def trunc(x): return (x.view(torch.int32) & -65536).view(torch.float32)
b1 = trunc(Wpad); r1 = Wpad - b1; b2 = trunc(r1); r2 = r1 - b2
assert torch.equal(b1 + b2 + r2, Wpad) and torch.equal(trunc(r2), r2)          # the three pieces are exact
planes = torch.stack([hi16(b1), hi16(b2), hi16(r2)]).contiguous().cuda()
bias = torch.zeros(112); bias[:NH] = b; bias = bias.cuda()
for R in (1000, 524288):
    X = torch.randn(R, F)
    if R == 1000: X *= torch.exp2(torch.randint(-12, 12, (R, F)).float())        # wide exponent range
    Xd = X.cuda(); H = torch.zeros(R, 112, device="cuda")
    args = (C.c_void_p * 6)(ptr(Xd), ptr(planes), ptr(bias), ival(R), ival(F), ptr(H))
    grid = min(256, (R + 255) // 256)
    for name in ("l1_full", "l1_nosplit", "l1_oneterm"):
        f = fn(name)
        hip.hipFuncSetAttribute(f, 8, LDS)                                      # hipFuncAttributeMaxDynamicSharedMemorySize
        rc = hip.hipModuleLaunchKernel(f, grid, 1, 1, 512, 1, 1, LDS, None, args, None)
        assert rc == 0, (name, rc)
        torch.cuda.synchronize()
        if name == "l1_full":
            ref = torch.relu(X.double() @ W.double().t() + b.double())
            got = H[:, :NH].cpu().double()
            scale = (X.double().abs() @ W.double().abs().t() + b.double().abs())
            f32 = torch.relu(X @ W.t() + b).double()
            print(f"R={R}: six-term bf16 vs float64: max |err| / (|x||w|) = {float(((got - ref).abs() / scale).max()):.2e};  torch CPU fp32: {float(((f32 - ref).abs() / scale).max()):.2e};  padding columns zero: {bool((H[:, NH:] == 0).all())}")
        if R > 100000:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for rep in range(3): hip.hipModuleLaunchKernel(f, grid, 1, 1, 512, 1, 1, LDS, None, args, None)
            e0.record()
            for rep in range(10): hip.hipModuleLaunchKernel(f, grid, 1, 1, 512, 1, 1, LDS, None, args, None)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            print(f"   {name:11s}: {us:7.1f} us  ({2.0 * R * F * NH / us / 1e6:6.1f} algorithmic TFLOP/s)")
    if R > 100000:
        from ptranking_amd import linear as LN
        w_d, b_d = W.cuda(), b.cuda()
        for rep in range(3): LN._fwd(Xd, F, w_d, b_d)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for rep in range(10): LN._fwd(Xd, F, w_d, b_d)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"   product fp32-MFMA linear kernel (no ReLU): {us:7.1f} us  ({2.0 * R * F * NH / us / 1e6:6.1f} TFLOP/s);  fp32 MFMA floor 7 x 9 x 4 MFMAs / 16 rows at 2.14 GHz: {524288 / 16 * 252 * 32 / 1024 / 2.14e3:6.1f} us")
