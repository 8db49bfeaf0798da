"""Issue cost of the ring kernels' VALU instruction classes on one MI355X (scratch/valu_rate/valu_rate.hip).
Build + run:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC scratch/valu_rate/valu_rate.hip -o scratch/valu_rate/libvalu_rate.so; python scratch/valu_rate/run.py"""
import ctypes as C, os, torch, numpy as np
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "libvalu_rate.so"))
lib.run_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
NAMES = ["v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_exp_f32", "v_rcp_f32", "v_log_f32", "v_fract_f32", "v_bfi_b32", "v_max_f32",
         "v_mov_b32_dpp wave_rol:1", "v_add_f32", "v_exp + 1 v_fma", "v_exp + 3 v_fma", "v_exp + 2 v_pk_fma"]
PER_ITER = [32] * 12 + [64, 128, 96]
sink = torch.zeros(1024 * 1024, device="cuda"); out = torch.zeros(1024 * 16, dtype=torch.int64, device="cuda")
ITERS = 2000
print(f"{'instruction class':28s} " + " ".join(f"{w} wave/SIMD".rjust(14) for w in (1, 2, 4)) + "   (SIMD-cycles per instruction; mixes: per GROUP of instructions)")
for mode, name in enumerate(NAMES):
    row = []
    for wps in (1, 2, 4):
        block = 256 * wps                       # 256 CUs x 1 block: wps waves on each of the 4 SIMDs
        for rep in range(2):
            assert lib.run_probe(mode, 256, block, ITERS, sink.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
            torch.cuda.synchronize()
        cyc = out[: 256 * block // 64].cpu().numpy().astype(np.float64)
        groups = ITERS * 32
        row.append(np.median(cyc) / (groups * wps))
    print(f"{name:28s} " + " ".join(f"{v:14.2f}" for v in row))
