"""Sustained shader clock of the MI355X under fp32-MFMA / VALU load (scratch/clock/clock_probe.hip, built by scratch/clock/build.sh)."""
import ctypes as C, os, sys, torch, numpy as np
here = os.path.dirname(os.path.abspath(__file__))
hip = C.CDLL("libamdhip64.so")
mod = C.c_void_p()
assert hip.hipModuleLoad(C.byref(mod), os.path.join(here, "clock_probe.hsaco").encode()) == 0
fns = {}
for m in range(7):
    fns[m] = C.c_void_p()
    assert hip.hipModuleGetFunction(C.byref(fns[m]), mod, f"probe{m}".encode()) == 0
grid = 256 * 2
sink = torch.zeros(1024 * 512 * 16 * 4, device="cuda"); out = torch.zeros(2 * 1024, dtype=torch.int64, device="cuda")
def run(mode, iters, grid=grid, block=512):
    fn = fns[mode]
    args = (C.c_void_p * 3)(C.cast(C.pointer(C.c_int(iters)), C.c_void_p),
                            C.cast(C.pointer(C.c_void_p(sink.data_ptr())), C.c_void_p), C.cast(C.pointer(C.c_void_p(out.data_ptr())), C.c_void_p))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = hip.hipModuleLaunchKernel(fn, grid, 1, 1, block, 1, 1, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream), args, None)
    assert rc == 0, rc
    e1.record(); torch.cuda.synchronize()
    o = out.cpu().numpy().reshape(-1, 2)[:grid].astype(np.float64)
    mhz = o[:, 0] / o[:, 1] * 100.0
    return e0.elapsed_time(e1), mhz
for mode, name, iters in ((0, "fp32 MFMA 16x16x4 back to back", 40000), (3, "same, all-zero operands", 40000), (4, "same, random mantissas", 40000), (2, "MFMA + one 1 KB store per 8", 40000), (1, "packed VALU fma", 80000)):
    for rep in range(3):
        ms, mhz = run(mode, iters)
    n_mfma = 8 * iters * 8 * grid                      # per wave 8 per iteration, 8 waves per block
    extra = f"  {n_mfma * 2048 / (ms * 1e-3) / 1e12:6.1f} TFLOP/s" if mode != 1 else ""
    print(f"{name:34s}: {ms:7.3f} ms  shader clock {mhz.mean():7.1f} MHz (min {mhz.min():.0f}, max {mhz.max():.0f}){extra}")

print("MFMA stream vs waves per SIMD:")
for g, blk in ((256, 256), (256, 512), (512, 512), (768, 512), (1024, 512)):
    for rep in range(2):
        ms, mhz = run(0, 40000, g, blk)
    n = 8 * 40000 * (blk // 64) * g
    print(f"  grid {g:5d} x {blk} threads ({g * blk // 64 / 1024:.0f} waves/SIMD): {ms:7.3f} ms  {n * 2048 / (ms * 1e-3) / 1e12:6.1f} TFLOP/s  clock {mhz.mean():.0f} MHz")

print("co-issue (512 blocks x 512 threads = 4 waves/SIMD):")
for rep in range(2):
    ms0, mhz0 = run(0, 20000, 512, 512)
for rep in range(2):
    ms5, mhz5 = run(5, 20000, 512, 512)
for rep in range(2):
    ms6, mhz6 = run(6, 20000, 512, 512)
n = 8 * 20000 * 8 * 512
print(f"  all 8 waves MFMA                         : {ms0:7.3f} ms  {n * 2048 / (ms0 * 1e-3) / 1e12:6.1f} TFLOP/s  {mhz0.mean():.0f} MHz")
print(f"  4 waves MFMA + 4 waves VALU (32 fma/iter): {ms5:7.3f} ms  {n / 2 * 2048 / (ms5 * 1e-3) / 1e12:6.1f} TFLOP/s from half the MFMA work  {mhz5.mean():.0f} MHz")
print(f"  every wave 8 MFMA + 16 VALU fma per iter : {ms6:7.3f} ms  {n * 2048 / (ms6 * 1e-3) / 1e12:6.1f} TFLOP/s  {mhz6.mean():.0f} MHz")
