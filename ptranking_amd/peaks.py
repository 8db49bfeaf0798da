"""MI355X (gfx950) hardware constants every roofline in this repository is priced against — ONE definition, imported by `bench.py`
and `profiles/prof_kernels.py` (the two files used different VALU issue constants; `tests/test_bench_contract.py`
fails when either stops importing these).  Source: /opt/skills/guides/MI355X_MICROARCH.md (chip table; "Per-instruction cycle
constants": `v_fma_f32 (wave64) 2 cyc (SIMD-32)`; transcendentals are quarter-rate: 8 cycles per wave64 instruction).
No torch import: usable from the CPU-only summarise steps.
"""
NUM_CUS = 256
NUM_SIMD = NUM_CUS * 4                    # 4 SIMD-32 units per CU
PEAK_CLOCK_HZ = 2.4e9
HBM_PEAK_GBPS = 8000.0                    # HBM3E spec peak (the guide measures 6.3 TB/s achievable with a float4 copy)
MFMA_F32_PEAK_TFLOPS = 157.3              # dense fp32-input MFMA peak = the fp32 vector peak
BF16_PEAK_TFLOPS = 2500.0                 # dense bf16 MFMA peak (AMD's 5 PF figure includes 2:1 sparsity)
BF16X6_PEAK_TFLOPS = BF16_PEAK_TFLOPS / 6  # effective fp32 peak of the bf16x6 formulation: six bf16 products per fp32 product
VALU_CYCLES_PER_INSTR = 2.0               # issue cycles of a plain wave64 VALU instruction on one SIMD-32
TRANS_CYCLES_PER_INSTR = 8.0              # v_exp_f32 / v_rcp_f32 / v_log_f32 (quarter rate)
VALU_PEAK_GINST = NUM_SIMD * PEAK_CLOCK_HZ / VALU_CYCLES_PER_INSTR / 1e9   # wave64 VALU instructions per ns, whole chip: 1228.8

# ---- the fused LambdaRank pair kernel's MINIMAL-op bound (bench.py `valu_roofline`, profiles/prof_kernels.py `pair_roofline`): 3
# transcendentals (exp2, rcp, log2) + 16 FMA-class scalar ops — ds, |ds|*c, 1+e, dG, dD, dG*dD, target select (sub, bfi, sub, add),
# max(log, clamp), loss fma, gradient factor (fract, bfi), two gradient fmas — all 16 packing two pairs per v_pk_* instruction:
# 8 packed + 3 transcendental issue slots per pair = 8 x 2 + 3 x 8 = 40 cycles.  A count of the arithmetic, not of our ISA.
RING_MIN_FMA_OPS_PER_PAIR = 16.0
RING_TRANS_PER_PAIR = 3.0
RING_MIN_ISSUE_CYCLES_PER_PAIR = RING_MIN_FMA_OPS_PER_PAIR / 2.0 * VALU_CYCLES_PER_INSTR + RING_TRANS_PER_PAIR * TRANS_CYCLES_PER_INSTR
RING_PAIR_PEAK_PER_S = NUM_SIMD * PEAK_CLOCK_HZ * 64.0 / RING_MIN_ISSUE_CYCLES_PER_PAIR      # 3.93e12 pairs/s
