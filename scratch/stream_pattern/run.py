"""r6: achievable X -> Y streaming rate of a kernel shaped like linear_fwd_x6t_kernel (scratch/stream_pattern/stream_pattern.hip): fragment pattern vs contiguous KBs,
8 / 16 waves per workgroup, 1 / 2 / 4 workgroups per CU.  262 144 rows x 136 floats in, the same out (285 MB per launch).
Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC scratch/stream_pattern/stream_pattern.hip -o scratch/stream_pattern/libstream_pattern.so"""
import ctypes as C, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libstream_pattern.so"))
lib.run_stream.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
R, N = 262144, 136
X = torch.randn(R + 64, N + 8, device="cuda"); Y = torch.empty(R + 64, N + 8, device="cuda")
X2 = torch.randn(R + 64, N, device="cuda").contiguous(); Y2 = torch.empty(R + 64, N, device="cuda")
ntiles = R // 32
st = torch.cuda.current_stream().cuda_stream
Xw = torch.randn(R + 64, 192, device="cuda"); Yw = torch.empty(R + 64, 192, device="cuda")        # ld = 192 (768-byte rows), 136 .. 191 = padding touched by modes 2 / 3
Xa = torch.randn(R + 64, 128, device="cuda"); Ya = torch.empty(R + 64, 128, device="cuda")        # 512-byte rows (aligned pieces), mode 0 over 8 column groups would need NLD = 16: use ld only
cases = [(0, "16 rows x 64 B, rows of 144 floats (576 B)", X, Y, N + 8, 2 * 18 * 1024 * (R // 32)),
         (1, "contiguous KBs (rows of 136 floats)", X2, Y2, N, 2 * 17 * 1024 * (R // 32)),
         (2, "8 rows x 128 B, rows of 192 floats (768 B)", Xw, Yw, 192, 2 * 20 * 1024 * (R // 32)),
         (3, "4 rows x 256 B, rows of 192 floats (768 B)", Xw, Yw, 192, 2 * 24 * 1024 * (R // 32)),
         (0, "16 rows x 64 B, rows of 192 floats (768 B)", Xw, Yw, 192, 2 * 18 * 1024 * (R // 32))]
cases += [(4, "16 rows x 64 B, rows of 128 floats (like for like)", Xa, Ya, 128, 2 * 16 * 1024 * (R // 32)),
          (5, "8 rows x 128 B, rows of 128 floats (like for like)", Xa, Ya, 128, 2 * 16 * 1024 * (R // 32)),
          (6, "4 rows x 256 B, rows of 128 floats (like for like)", Xa, Ya, 128, 2 * 16 * 1024 * (R // 32))]
for mode, name, xx, yy, ldx, nbytes in cases:
    for waves, wg_per_cu in ((8, 1), (16, 2)):
        grid = 256 * wg_per_cu
        for _ in range(3):
            assert lib.run_stream(mode, waves, grid, xx.data_ptr(), yy.data_ptr(), ldx, ldx, ntiles, 32, st) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.run_stream(mode, waves, grid, xx.data_ptr(), yy.data_ptr(), ldx, ldx, ntiles, 32, st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"{name:46s} {waves:2d} waves/WG x {wg_per_cu} WG/CU: {us:7.1f} us  {nbytes / us / 1e6:5.2f} TB/s of touched bytes", flush=True)
