"""4x4x1 (16 blocks) fp32 MFMA: lane layout and issue cost (scratch/mfma4/probe.hip -> probe.hsaco)."""
import ctypes as C, os, torch, numpy as np
here = os.path.dirname(os.path.abspath(__file__))
hip = C.CDLL("libamdhip64.so")
mod = C.c_void_p()
assert hip.hipModuleLoad(C.byref(mod), os.path.join(here, "probe.hsaco").encode()) == 0
def fn(name):
    f = C.c_void_p(); assert hip.hipModuleGetFunction(C.byref(f), mod, name.encode()) == 0; return f
out = torch.zeros(256, device="cuda")
args = (C.c_void_p * 1)(C.cast(C.pointer(C.c_void_p(out.data_ptr())), C.c_void_p))
assert hip.hipModuleLaunchKernel(fn("layout"), 1, 1, 1, 64, 1, 1, 0, None, args, None) == 0
torch.cuda.synchronize()
d = out.cpu().numpy().reshape(64, 4)
print("D[lane][reg] = a(lane x) * b(lane y): (x, y)")
for l in range(64):
    row = []
    for r in range(4):
        v = d[l, r]; y = int(round(v / 1000)) // 1  # v = (x+1) * 1000 (y+1)
        # factor: v/1000 = (x+1)(y+1); search
        found = [(x, yy) for x in range(64) for yy in range(64) if abs((x + 1) * 1000.0 * (yy + 1) - v) < 0.5 and x // 4 == yy // 4]
        row.append(found)
    if l < 12 or l % 16 == 0: print(l, row)
sink = torch.zeros(4, device="cuda"); o = torch.zeros(1024, dtype=torch.int64, device="cuda")
for name, what, n in (("t0", "8 x 16x16x4", 8), ("t1", "8 x 4x4x1 independent", 8), ("t2", "6 x 16x16x4 + 2 x 4x4x1", 8), ("t3", "2 chains of 4 dependent 4x4x1", 8), ("t4", "8 dependent 4x4x1", 8)):
    iters = 20000
    a = (C.c_void_p * 3)(C.cast(C.pointer(C.c_int(iters)), C.c_void_p), C.cast(C.pointer(C.c_void_p(sink.data_ptr())), C.c_void_p), C.cast(C.pointer(C.c_void_p(o.data_ptr())), C.c_void_p))
    for rep in range(2):
        assert hip.hipModuleLaunchKernel(fn(name), 256, 1, 1, 256, 1, 1, 0, None, a, None) == 0
        torch.cuda.synchronize()
    cyc = o.cpu().numpy()[:256].mean() / iters
    print(f"{what:32s}: {cyc:7.1f} cycles per iteration (1 wave / SIMD)")
