"""fp32 products from bf16 MFMAs (scratch/bf16x6/probe.hip -> probe.hsaco): accuracy of the 6-term / 3-term split vs the fp32 MFMA, issue
cost and sustained clock of v_mfma_f32_16x16x32_bf16 next to v_mfma_f32_16x16x4_f32."""
import ctypes as C, os, torch, numpy as np
here = os.path.dirname(os.path.abspath(__file__))
hip = C.CDLL("libamdhip64.so")
mod = C.c_void_p()
assert hip.hipModuleLoad(C.byref(mod), os.path.join(here, "probe.hsaco").encode()) == 0
def fn(name):
    f = C.c_void_p(); assert hip.hipModuleGetFunction(C.byref(f), mod, name.encode()) == 0; return f
def ptr(t): return C.cast(C.pointer(C.c_void_p(t.data_ptr())), C.c_void_p)
torch.manual_seed(0)
for K, kind in ((128, "randn"), (1024, "randn"), (128, "wide range"), (128, "positive")):
    A = torch.randn(16, K); B = torch.randn(K, 16)
    if kind == "wide range": A = A * torch.exp2(torch.randint(-20, 20, (16, K)).float()); B = B * torch.exp2(torch.randint(-20, 20, (K, 16)).float())
    if kind == "positive": A = A.abs(); B = B.abs()
    out = torch.zeros(3, 16, 16, device="cuda")
    Ad, Bd = A.cuda(), B.cuda()
    args = (C.c_void_p * 4)(ptr(Ad), ptr(Bd), C.cast(C.pointer(C.c_int(K)), C.c_void_p), ptr(out))
    assert hip.hipModuleLaunchKernel(fn("accuracy"), 1, 1, 1, 64, 1, 1, 0, None, args, None) == 0
    torch.cuda.synchronize()
    ref = A.double() @ B.double()
    scale = (A.double().abs() @ B.double().abs())
    o = out.cpu().double()
    f32 = (A @ B).double()
    for name, d in (("fp32 MFMA", o[0]), ("6 bf16 terms", o[1]), ("3 bf16 terms", o[2]), ("torch CPU fp32", f32)):
        err = (d - ref).abs()
        print(f"K={K:5d} {kind:11s} {name:15s}: max |err| / (|A||B|) = {float((err / scale).max()):.2e}   max |err| / max|ref| = {float(err.max() / ref.abs().max()):.2e}")
sink = torch.zeros(4, device="cuda"); o = torch.zeros(8192, dtype=torch.int64, device="cuda")
for wpb in (4, 8):
    for name, what in (("r0", "8 x 16x16x4 f32"), ("r1", "8 x 16x16x32 bf16 independent"), ("r2", "8 x 16x16x32 bf16 dependent"), ("r3", "8 x (bf16 MFMA + and/sub/fma VALU)")):
        iters = 100000
        a = (C.c_void_p * 3)(C.cast(C.pointer(C.c_int(iters)), C.c_void_p), ptr(sink), ptr(o))
        for rep in range(2):
            o.zero_()
            assert hip.hipModuleLaunchKernel(fn(name), 256, 1, 1, 64 * wpb, 1, 1, 0, None, a, None) == 0
            torch.cuda.synchronize()
        r = o.cpu().numpy()
        cyc = r[:256].mean() / iters; us = r[4096:4096 + 256].mean() / 100.0
        print(f"{wpb // 4} wave(s)/SIMD, all 256 CUs  {what:36s}: {cyc / 8:6.1f} shader cycles per MFMA per wave, {r[:256].mean() / us / 1e3:5.2f} GHz sustained over {us / 1e3:.1f} ms")

print("VALU beside fp32 MFMAs (per v_mfma_f32_16x16x4_f32: n fma, or n x (xor-shift + 32-bit multiply) = 3n instructions)")
for wpb in (4, 8, 16):
    for name, what in (("v0", "0"), ("v2", "2 fma"), ("v4", "4 fma"), ("v6", "6 fma"), ("v8", "8 fma"), ("v12", "12 fma"), ("m1", "1 hash step"), ("m2", "2 hash steps"), ("m4", "4 hash steps")):
        iters = 30000
        a = (C.c_void_p * 3)(C.cast(C.pointer(C.c_int(iters)), C.c_void_p), ptr(sink), ptr(o))
        for rep in range(2):
            o.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert hip.hipModuleLaunchKernel(fn(name), 256, 1, 1, 64 * wpb, 1, 1, 0, None, a, None) == 0
            e1.record()
            torch.cuda.synchronize()
        r = o.cpu().numpy()
        cyc = r[:256].mean() / iters / 8          # slowest wave of each workgroup
        ms = e0.elapsed_time(e1)
        print(f"{wpb // 4} wave(s)/SIMD  {what:14s}: {cyc / (wpb // 4):6.1f} SIMD cycles per MFMA (slowest wave);  kernel {ms:7.2f} ms = {ms * 1e6 / (iters * 8 * (wpb // 4)):6.2f} ns per MFMA per SIMD")

print("which VALU classes cost matrix-pipe time: 4 instructions of one class per v_mfma_f32_16x16x4_f32, 4 waves per SIMD (32.0 = free)")
for name, what in (("c0", "4 x v_fma_f32"), ("c1", "4 x v_xor_b32"), ("c2", "2 x (v_lshrrev_b32 + v_or_b32)"), ("c3", "4 x v_mul_lo_u32"), ("c4", "2 x (v_cmp_ge_u32 + v_cndmask_b32)"),
                   ("c5", "4 x v_max_f32"), ("c6", "4 x v_mul_f32"), ("c7", "4 x v_add_u32")):
    iters = 30000
    a = (C.c_void_p * 3)(C.cast(C.pointer(C.c_int(iters)), C.c_void_p), ptr(sink), ptr(o))
    for rep in range(2):
        o.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert hip.hipModuleLaunchKernel(fn(name), 256, 1, 1, 1024, 1, 1, 0, None, a, None) == 0
        e1.record(); torch.cuda.synchronize()
    cyc = o.cpu().numpy()[:256].mean() / iters / 8 / 4
    print(f"   {what:36s}: {cyc:6.1f} SIMD cycles per MFMA (slowest wave); kernel {e0.elapsed_time(e1):6.2f} ms")
