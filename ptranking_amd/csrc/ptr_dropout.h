// Counter-based dropout bits and the fp32 MFMA vector type, shared by the fused scorers (scorer.hip, listsf.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptr {

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
constexpr uint32_t kDropRowMul = 0x9E3779B1u, kDropFgMul = 0x85EBCA77u, kDropSiteMul = 0xC2B2AE3Du;
// the generator proper: ks = row * kDropRowMul + fg * kDropFgMul + site * kDropSiteMul + seed_lo (callers that keep the per-lane part of
// the key in a register and add a scalar per feature group — scorer_x6.hip — enter here)
__device__ __forceinline__ void drop_bits_key(uint32_t ks, uint32_t seed_hi, uint32_t &w0, uint32_t &w1) {
    w0 = lowbias32(ks);
#ifdef PTR_DROP_FULL_HASH
    w1 = lowbias32(w0 ^ seed_hi ^ 0x68E31DA4u);
#else
    uint32_t t = w0 ^ seed_hi ^ 0x68E31DA4u;
    t ^= t >> 15; t *= 0x2C1B3C6Du; t ^= t >> 13;
    w1 = t;
#endif
}
// 64 random bits for features [4*fg, 4*fg+3] of `row` at dropout site `site` (site l = the Dropout in front of hidden layer l)
__device__ __forceinline__ void drop_bits(uint32_t seed_lo, uint32_t seed_hi, int site, int row, int fg, uint32_t &w0, uint32_t &w1) {
    // A VALU instruction takes its cycles away from the matrix pipe of its SIMD (no MFMA / VALU overlap: scratch/clock), and 32-bit
    // integer multiplies are quarter rate on CDNA: the row term is loop invariant for a lane (hoisted by the compiler), the
    // first word gets the full two-multiply finaliser, the second word one more multiply-xorshift round on top of it
    // (PTR_DROP_FULL_HASH restores the round-1 generator: a second full finaliser)
    // seed_lo enters ADDITIVELY: a replica that owns rows [row0, row0 + n) of a global batch passes seed_lo + row0 * 0x9E3779B1 and draws
    // exactly the masks the single-device run draws for those rows (ptranking_amd/dp.py fold_row_offset) — data-parallel replicas never
    // share masks, and N ranks x B/N queries reproduce one rank x B
    const uint32_t key = (uint32_t)row * kDropRowMul + ((uint32_t)fg * kDropFgMul + (uint32_t)site * kDropSiteMul);
    drop_bits_key(key + seed_lo, seed_hi, w0, w1);
}
__device__ __forceinline__ f32x4 drop4(f32x4 v, uint32_t w0, uint32_t w1, uint32_t thr, float scale) {
    // multiplicative masks on purpose: with a select hipcc sinks the producing global load under the predicate and
    // serialises it behind a vmcnt(0); x * 0.0f cannot be folded without fast-math, so the load stays unconditional
    f32x4 o;
    o[0] = v[0] * ((w0 & 0xFFFFu) >= thr ? scale : 0.0f);
    o[1] = v[1] * ((w0 >> 16) >= thr ? scale : 0.0f);
    o[2] = v[2] * ((w1 & 0xFFFFu) >= thr ? scale : 0.0f);
    o[3] = v[3] * ((w1 >> 16) >= thr ? scale : 0.0f);
    return o;
}
__device__ __forceinline__ bool drop_keep1(uint32_t seed_lo, uint32_t seed_hi, int site, int row, int k, uint32_t thr) {
    uint32_t w0, w1;
    drop_bits(seed_lo, seed_hi, site, row, k >> 2, w0, w1);
    const uint32_t w = (k & 2) ? w1 : w0;
    return ((k & 1) ? (w >> 16) : (w & 0xFFFFu)) >= thr;
}

__device__ __forceinline__ uint32_t drop_thr(float p) { return (uint32_t)(p * 65536.0f + 0.5f); }

}  // namespace ptr
