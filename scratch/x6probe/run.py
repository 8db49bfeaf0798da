"""r4 probes: VALU beside v_mfma_f32_16x16x32_bf16 (1 / 2 waves per SIMD, all CUs) and the delivery pattern / bank behaviour of
ds_read_b64_tr_b16.  Build: hipcc -O3 --offload-arch=gfx950 --genco probe.hip -o probe.hsaco"""
import ctypes as C, os, torch, numpy as np
here = os.path.dirname(os.path.abspath(__file__))
hip = C.CDLL("libamdhip64.so")
mod = C.c_void_p()
assert hip.hipModuleLoad(C.byref(mod), os.path.join(here, "probe.hsaco").encode()) == 0
def fn(name):
    f = C.c_void_p(); assert hip.hipModuleGetFunction(C.byref(f), mod, name.encode()) == 0; return f
def ptr(t): return C.cast(C.pointer(C.c_void_p(t.data_ptr())), C.c_void_p)
def ci(v): return C.cast(C.pointer(C.c_int(v)), C.c_void_p)
sink = torch.zeros(4, device="cuda"); o = torch.zeros(8192, dtype=torch.int64, device="cuda")
import sys
DEP_ONLY = "--dep-only" in sys.argv
print("VALU beside v_mfma_f32_16x16x32_bf16: SIMD cycles per MFMA (slowest wave of a workgroup / waves per SIMD); 16-17 = hidden")
for wpb in (() if DEP_ONLY else (4, 8)):
    for name, what in (("f0", "none"), ("f1", "1 v_fma_f32"), ("f2", "2 v_fma_f32"), ("f3", "3 v_fma_f32"), ("f4", "4 v_fma_f32"), ("f6", "6 v_fma_f32"),
                       ("a2", "2 v_and_b32"), ("a3", "3 v_and_b32"), ("p2", "2 v_perm_b32"), ("p3", "3 v_perm_b32"), ("m1", "1 v_mul_lo_u32"), ("m2", "2 v_mul_lo_u32"),
                       ("s1", "1 cmp+cndmask"), ("s2", "2 cmp+cndmask"), ("u3", "3 v_sub_f32"), ("h3", "3 v_lshrrev_b32")):
        iters = 30000
        a = (C.c_void_p * 3)(ci(iters), ptr(sink), ptr(o))
        for rep in range(2):
            o.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert hip.hipModuleLaunchKernel(fn(name), 256, 1, 1, 64 * wpb, 1, 1, 0, None, a, None) == 0
            e1.record(); torch.cuda.synchronize()
        cyc = o.cpu().numpy()[:256].mean() / iters / 8 / (wpb // 4)
        ms = e0.elapsed_time(e1)
        print(f"  {wpb // 4} wave(s)/SIMD  {what:16s}: {cyc:6.1f} cycles;  kernel {ms:6.2f} ms = {ms * 1e6 / (iters * 8 * (wpb // 4)):6.2f} ns per MFMA per SIMD", flush=True)

print("dependent accumulators: SIMD cycles per v_mfma_f32_16x16x32_bf16 with NA accumulators visited round-robin")
for wpb in (4, 8):
    for name, what in (("d1", "1 (back to back on one accumulator)"), ("d2", "2 (alternating)"), ("d4", "4"), ("d8", "8")):
        iters = 30000
        a = (C.c_void_p * 3)(ci(iters), ptr(sink), ptr(o))
        for rep in range(2):
            o.zero_()
            assert hip.hipModuleLaunchKernel(fn(name), 256, 1, 1, 64 * wpb, 1, 1, 0, None, a, None) == 0
            torch.cuda.synchronize()
        cyc = o.cpu().numpy()[:256].mean() / iters / 8 / (wpb // 4)
        print(f"  {wpb // 4} wave(s)/SIMD  NA = {what:36s}: {cyc:6.1f} cycles", flush=True)
if DEP_ONLY: sys.exit(0)
print("ds_read_b64_tr_b16 delivery: lane l passes byte address addr[l]; LDS halfword i holds i")
out = torch.zeros(256, dtype=torch.int32, device="cuda")
def tr(addr):
    ad = torch.tensor(addr, dtype=torch.int32, device="cuda")
    a = (C.c_void_p * 2)(ptr(ad), ptr(out))
    assert hip.hipModuleLaunchKernel(fn("trread"), 1, 1, 1, 64, 1, 1, 0, None, a, None) == 0
    torch.cuda.synchronize()
    return out.cpu().numpy().reshape(64, 4).copy()
r = tr([8 * l for l in range(64)])
print(" addr = 8*l:  lane 0", r[0], " lane 1", r[1], " lane 4", r[4], " lane 15", r[15], " lane 16", r[16], " lane 63", r[63])
guide = np.array([[(l & 15) + j * 16 + (l >> 4) * 64 for j in range(4)] for l in range(64)])
print("   == guide formula lds[(l&15) + j*16 + (l>>4)*64]:", bool((r == guide).all()))
for stride in (224, 256, 272, 288):
    # row-major [doc][feature] bf16 image, row stride `stride` bytes; 16-lane group g covers docs 4g..4g+3, features 0..15:
    # lane t of the group passes the address of chunk (row 4g + t//4, features 4(t%4)..+3)
    addr = [((4 * (l >> 4)) + (l & 15) // 4) * stride + ((l & 15) % 4) * 8 for l in range(64)]
    r = tr(addr)
    want = np.array([[(4 * (l >> 4) + j) * (stride // 2) + (l & 15) for j in range(4)] for l in range(64)])
    print(f" [doc][feature] stride {stride}: lane (i, g) receives docs 4g..4g+3 of feature i:", bool((r == want).all()), " lane 5:", r[5], "want", want[5])
    addr = [((4 * (l >> 4)) + (l & 15) % 4) * stride + ((l & 15) // 4) * 8 for l in range(64)]
    r2 = tr(addr)
    print("     alt assignment (row 4g + t%4, chunk t//4):", bool((r2 == want).all()), " lane 5:", r2[5])

print("ds_read_b64_tr_b16 time: 8 waves, 8 reads / iteration, cycles per wave-instruction per CU (2 = conflict free)")
o2 = torch.zeros(8192, dtype=torch.int64, device="cuda"); sk = torch.zeros(4, dtype=torch.int32, device="cuda")
assert hip.hipFuncSetAttribute(fn("trtime"), 8, 65536) in (0, 1) or True
for stride in (224, 232, 240, 256, 272, 288, 64, 32):
    addr = [((4 * (l >> 4)) + (l & 15) // 4) * stride + ((l & 15) % 4) * 8 for l in range(64)]
    ad = torch.tensor(addr, dtype=torch.int32, device="cuda")
    iters = 20000
    a = (C.c_void_p * 4)(ptr(ad), ci(iters), ptr(o2), ptr(sk))
    for rep in range(2):
        o2.zero_()
        rc = hip.hipModuleLaunchKernel(fn("trtime"), 256, 1, 1, 512, 1, 1, 65536, None, a, None)
        assert rc == 0, rc
        torch.cuda.synchronize()
    cyc = o2.cpu().numpy()[:256].mean() / iters / 8 / 8
    print(f"  stride {stride:4d} B: {cyc:5.2f} cycles per wave-instruction per CU (8 waves issuing)")
