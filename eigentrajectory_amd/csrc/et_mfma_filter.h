// et_mfma_filter.h -- the pieces of the matrix-core label filter (csrc/et_kmeans.hip: "Lloyd half-step for iterations >= 1",
// where the bounds are derived) that the reference-order Lloyd kernel (csrc/et_kmeans_reforder.hip) shares with it: the f16
// (hi, lo) split, the top-2 of an accumulator, v_med3-based max / min on raw MFMA results.
#pragma once

#include "et_common.h"

namespace et {

__device__ __forceinline__ int exponent_above(double m) {  // smallest E with m < 2^E; 0 for m == 0
    if (!(m > 0.0)) return 0;
    const unsigned long long b = (unsigned long long)__double_as_longlong(m);
    return (int)((b >> 52) & 0x7ff) - 1022;
}


typedef float f32x16 __attribute__((ext_vector_type(16)));


typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// (a, b) * sg (a power of two) -> packed f16 {hi(a), hi(b)} and {lo(a), lo(b)} with sg a ~ hi + lo, round to nearest:
// |sg a - hi - lo| <= 2^-22 |sg a| + 2^-25 (inside the 2^-20 relative + 2^-24 absolute the bounds below are derived
// with).  Written as fused multiply-adds rounded to f16, which the compiler selects as v_fma_mixlo/mixhi_f16: they scale
// and round in one step (a sg is exact in fp32, sg being a power of two) and evaluate the residual a sg - hi exactly (it
// has at most 13 significant bits) before rounding it to f16 -- five instructions per pair, no separate scaling
// multiplies, no pack instructions.  NOT inline assembly (the four-instruction form rounds of 2-5 used): the hazard
// recogniser cannot see registers an asm statement writes, and next to matrix instructions the allocator handed the asm
// the B operands of MFMAs that were still reading them (round 5, reference-order kernel: 0.5 % wrong labels in one
// build, none in another).  Every site uses this form, whether or not an MFMA is in flight.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_f16(float a, float b, float sg, unsigned &hi, unsigned &lo) {
    f16x2_t h, l;
    h.x = (_Float16)__builtin_fmaf(a, sg, 0.0f);
    h.y = (_Float16)__builtin_fmaf(b, sg, 0.0f);
    l.x = (_Float16)__builtin_fmaf(a, sg, -(float)h.x);
    l.y = (_Float16)__builtin_fmaf(b, sg, -(float)h.y);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

// max(|a|, |b|, |c|) in one instruction (from fmaxf(fmaxf(fabsf ...)) the compiler canonicalises two of the operands
// with a v_max x,x each)
__device__ __forceinline__ float max3_abs(float a, float b, float c) {
    float m;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(m) : "v"(a), "v"(b), "v"(c));
    return m;
}

// max / min as v_med3_f32 against +-inf: fmaxf on a raw MFMA result would first be "canonicalised" by a
// v_max x,x (the compiler cannot prove it quiet), one extra VALU op per value.  (Inline asm is not an option:
// the compiler does not insert the MFMA -> VALU wait states in front of instructions it cannot see.)
__device__ __forceinline__ float opaque_inf() {  // +inf the optimiser cannot see through (no instruction is emitted)
    unsigned u = 0x7f800000u;
    asm volatile("" : "+s"(u));
    return __uint_as_float(u);
}
__device__ __forceinline__ float vmax(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, opaque_inf()); }
__device__ __forceinline__ float vmin(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, -opaque_inf()); }
__device__ __forceinline__ float vmed3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

// largest and second largest (as a multiset) of acc[0..NREGS), NREGS >= 3: (max3, med3) per triple; the triples' pairs
// are merged two at a time into the running pair -- largest = max3 of the three maxima, second = the largest of their
// median and the three seconds (four instructions for two triples) --, a last odd pair with three instructions, left-over
// values with two each.  Ten values: 12 instructions (the pairwise merge this replaces: 14).
template <int NREGS>
__device__ __forceinline__ void top2(const f32x16 &acc, float &b, float &s) {
    constexpr int T = NREGS / 3;
    b = __builtin_fmaxf(__builtin_fmaxf(acc[0], acc[1]), acc[2]);
    s = vmed3(acc[0], acc[1], acc[2]);
    int i = 1;
#pragma unroll
    for (; i + 1 < T; i += 2) {
        const int r = 3 * i;
        const float m1 = __builtin_fmaxf(__builtin_fmaxf(acc[r], acc[r + 1]), acc[r + 2]);
        const float t1 = vmed3(acc[r], acc[r + 1], acc[r + 2]);
        const float m2 = __builtin_fmaxf(__builtin_fmaxf(acc[r + 3], acc[r + 4]), acc[r + 5]);
        const float t2 = vmed3(acc[r + 3], acc[r + 4], acc[r + 5]);
        const float mid = vmed3(b, m1, m2);
        s = __builtin_fmaxf(__builtin_fmaxf(mid, s), vmax(t1, t2));
        b = __builtin_fmaxf(__builtin_fmaxf(b, m1), m2);
    }
    if (i < T) {
        const int r = 3 * i;
        const float gs = vmed3(acc[r], acc[r + 1], acc[r + 2]);
        const float gm = __builtin_fmaxf(__builtin_fmaxf(acc[r], acc[r + 1]), acc[r + 2]);
        s = vmed3(b, gm, vmax(s, gs));
        b = vmax(b, gm);
    }
#pragma unroll
    for (int r = 3 * T; r < NREGS; ++r) {
        s = vmed3(b, s, acc[r]);
        b = vmax(b, acc[r]);
    }
}


}  // namespace et
