// et_kmeans_packed.inl -- part of csrc/et_kmeans.hip (ONE translation unit: this file is #included there, in order, and is not
// compiled on its own): trace-less Lloyd iterations on the packed f16 copy of the points: PackedHeader, kmeans_pack_kernel, the per-launch tables, packed_assign_body.
// clang-format off: the fragment starts and ends at namespace scope of whatever the including file has open.
// The filter above reads 24 B per point and iteration to certify that a label did not change, and an iteration takes as
// long as the memory side needs to stream them (HISTORY.md 3.2).  The certification does not need the exact coordinates:
// kmeans_pack_kernel writes, once per fit,
//   xh   three rows of N dwords: the coordinate pairs (0,1), (2,3), (4,5) of  s (x - mu)  rounded to f16 (nearest);
//        mu = the mean of 1024 evenly spaced points (any vector would do: arg-max_j -|x - c_j|^2 does not depend on the
//        origin), s = the power of two that brings every |s (x - mu)| below 16
//   rr   N f16: an upper bound R of  s ||x - mu||
//   xa   32 B per point: the exact coordinates of a point side by side (where: xa_index), for the few points the test cannot decide
// = 14 B per point and iteration instead of 24, and the 1-3 % of undecided points cost one 64-B sector each instead of
// six.  The rounding of x is now by far the largest error of the matrix-core estimate, so the bounds are re-derived
// (scaled units; p = s (x - mu), q_j = s (c_j - mu) exact, R >= ||p||, Q_j >= ||q_j||; r = s ||x|| <= R + m with
// m >= s ||mu||, C_j = s ||c_j||, M_j = m + C_j;  G_j = 2 p.q_j - |q_j|^2, and G_j - G_l = s^2 (Y_j - Y_l) in exact
// arithmetic whatever mu is):
//   |xh_i - p_i|  <= (2^-11 + 2^-23) |p_i| + 2^-25          (x - mu in fp32, then f16 to nearest / denormal grid)
//   the MFMA's  t'_j = sum_i xh_i (ch + cl)_ji - |q_j|^2  (2 q_j split into f16 hi + lo as before, fp32 accumulation
//   of 16 terms):   |t'_j - G_j| <= E2 = 2^-9.99 R Q_j + 2^-17.5 (R + Q_j)^2 + 2^-21 (R + Q_j) + 2^-34
//   the reference's fp32 chain (kmeans.py:71-74):  |s^2 (Y_j + |x|^2) - (G_j + |p|^2 ... )| -- only differences
//   matter --  is within E1_j = 2^-20.99 (R + M_j)^2 of the exact value (the 2^-21 (r + C_j)^2 of the filter above)
//   so  s^2 Y_j - const <= u_j := t'_j + E2_j + E1_j,  and  u_j - epsR(R)  is linear in (R, 1) per cluster: the slope
//   rides in a k-slot against R, the constant is folded into the -|q_j|^2 slots (rounded up).
//   The old label l:  w' = sum_i xh_i (2 s c~_li) - s^2 |c~_l|^2  as an fp32 chain on the f16 values (v_fma_mix_f32),
//   G_l >= w' - Ew_l,  Ew_l = 2^-9.99 R Q_l + 2^-20 (R + Q_l)^2 + 2^-22 Q_l + 2^-40.
//   keep  <=>  w' - second > epsR(R) + Ew_l + E1_l (+ the rounding of the comparison):  then l owns the largest u (were it
//   not, u_l <= second would give w' <= second + epsR + Ew_l) and every other cluster j has  s^2 Y_j - const <= second +
//   epsR < w' - Ew_l - E1_l <= s^2 Y_l - const:  l is the reference's arg-max, strictly.
// Everything else -- the queue, the exact scan of the queued points (now on coordinates fetched from xa), the
// incremental integer sums -- is the filter's; labels, sums and iteration counts stay bit-identical.  Falls back to the
// fp32 filter for an iteration whose centroids leave the packed range (|s (c - mu)| >= 31: cannot happen for means of
// the points, can for caller-provided initial centroids) or when the scale is out of range.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kKmThreads) void kmeans_pack_kernel(const float *__restrict__ X, int64_t N,
                                                                 const et_kmeans_state *__restrict__ state,
                                                                 PackedHeader *__restrict__ hdr, unsigned *__restrict__ xh,
                                                                 unsigned short *__restrict__ rr, float4 *__restrict__ xa) {
    // (stand-alone form, ET_KMEANS_PACK_FUSED=0: by default the exact first iteration of the fit writes the copy)
    constexpr int d = 6;
    PackOut po;
    po.xh = xh;
    po.rr = rr;
    po.xa = xa;
    if (!packed_header(X, N, state, hdr, po.mu, po.s)) return;
    const int tid = threadIdx.x;
    const int64_t n_quads = N / 4;  // N % 4 == 0 (the caller's vec_ok)
    for (int64_t g = (int64_t)blockIdx.x * kKmThreads + tid; g < n_quads; g += (int64_t)gridDim.x * kKmThreads) {
        const int64_t n = 4 * g;
        float x[4][d];
#pragma unroll
        for (int i = 0; i < d; ++i) {
            const float4 v = *reinterpret_cast<const float4 *>(X + (int64_t)i * N + n);
            x[0][i] = v.x;
            x[1][i] = v.y;
            x[2][i] = v.z;
            x[3][i] = v.w;
        }
        pack_quad(x, n, N, po);
    }
}

// a += f16(w.lo or w.hi) * b: the compiler folds the (exact) conversion into one v_fma_mix_f32.  Compiler-visible on
// purpose: these instructions sit between matrix instructions, and the hazard recogniser does not look inside inline
// assembly (HISTORY.md 3.8: an asm helper's output once landed in a register an earlier v_mfma was still reading).
__device__ __forceinline__ float fma_mix_lo(unsigned w, float b, float a) {
    return fmaf((float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu)), b, a);
}
__device__ __forceinline__ float fma_mix_hi(unsigned w, float b, float a) {
    return fmaf((float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)), b, a);
}

constexpr int kPkQueue = 2 * kFilterSlots;  // per wavefront: point index, old label
constexpr int kPkRow = 12;                  // floats per cluster in the label table

#ifndef ET_PK_ACC_COPIES
#define ET_PK_ACC_COPIES 4
#endif
constexpr int kPkAccCopies = ET_PK_ACC_COPIES;  // copies of the workgroup's accumulators in LDS (packed_drain); a power of two
constexpr int kPkAccPitch = 228;   // int64 per copy: >= d K + K + 2 = 226 for K = 32; 456 words = 8 mod 64: eight different banks

// full exact scan of `cnt` (<= 64) queued points, one per lane, on coordinates fetched from the side-by-side copy
__device__ __forceinline__ void packed_drain(const unsigned *q, int cnt, int K, const float *sC, const float4 *__restrict__ xa,
                                             uint8_t *__restrict__ labels, long long *sAcc, int frac, int lane) {
    constexpr int d = 6;
    // At most 32 queued points (the usual case of a wavefront's LAST drain, which sits on the launch's tail with nothing to
    // hide behind): lanes e and e + 32 both fetch point e and scan about half of the centroids each -- the first kh (a multiple
    // of four) and the rest --, then the lower lane merges: the later range wins only by the scan's own comparison
    // (strictly larger, or NaN against non-NaN), i.e. the result is the one of the scan over all K in order.
    const bool halves = __builtin_amdgcn_readfirstlane(cnt) <= 32 && K >= 8;  // (wave-uniform, and known to be)
    const int e = halves ? (lane & 31) : lane;
    const bool mine = e < cnt;
    const int ec = mine ? e : 0;
    const int64_t n = (int64_t)q[ec];
    const int old = (int)q[kFilterSlots + ec];
    const int64_t ia = xa_index(n);
    const float4 a = xa[ia], b = xa[ia + 1];
    const float x[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
    int lb;
    float best;
    const bool up = halves && lane >= 32;
    const int kh = ((K + 4) / 8) * 4;  // K = 20: twelve and eight
    best_centroid6_drain(x, sC, (halves && !up) ? kh : K, lb, best, up ? kh : 0);
    if (halves) {
        const auto rb = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
        const auto rl = __builtin_amdgcn_permlane32_swap((unsigned)lb, (unsigned)lb, false, false);
        const float ub = __uint_as_float(rb[1]);  // (second result, lower lanes: the upper partner's value)
        const bool take = gt_nanmax(ub, best);
        lb = take ? (int)rl[1] : lb;
    }
    if (mine && !up && lb != old) {
        labels[n] = (uint8_t)lb;
        // kPkAccCopies copies of the accumulators, a lane adds onto copy lane % kPkAccCopies: the points that change in
        // one iteration move between a handful of clusters, so the 64 lanes of a drain hit a few addresses each, and LDS
        // atomics of one instruction on the same address are executed one after the other.  Same-box rocprofv3 averages
        // over the bench's 100 iterations: 1 copy 40.3 / 40.5 us, 2: 39.8 / 40.2, 4: 39.7 / 39.8, 8: 39.7 / 40.1 (the
        // iterations in which 3 % of the points move gain 4 us, the quiet ones pay 0.5 us for clearing and folding)
        long long *acc = sAcc + (lane & (kPkAccCopies - 1)) * kPkAccPitch;
        atomicAdd(reinterpret_cast<unsigned long long *>(&acc[d * K + lb]), 1ull);
        atomicAdd(reinterpret_cast<unsigned long long *>(&acc[d * K + old]), ~0ull);
#pragma unroll
        for (int i = 0; i < d; ++i) {
            const unsigned long long f = (unsigned long long)to_fixed(x[i], frac);
            atomicAdd(reinterpret_cast<unsigned long long *>(&acc[i * K + lb]), f);
            atomicAdd(reinterpret_cast<unsigned long long *>(&acc[i * K + old]), 0ull - f);
        }
    }
}

// Dual form of a pass's loads: BOTH lanes of a column request the rows of the column's point in the lower 128-point block
// (-> the B operand of tile L) and in the upper block (tile U) -- the two half-waves ask for the same addresses, the memory
// side sees the bytes once -- instead of exchanging their own rows with v_permlane32_swap (2 issue slots + 2 copies per
// dword).  A lane's own point is the lower block's for half 0, the upper block's for half 1.  Buffer loads: a pass index
// past the end (g < 0: offset 0xfffffff0) or rows past N are out of range, return zeros and cost no traffic, so the request
// needs no branch around it and the two register sets of the loop (unrolled by two: no copies) are waited for by count.
struct PkRows {
    u32x4 vL[3], vU[3];
    unsigned rL[2], rU[2];
    unsigned lp;
};
struct PkSrc {
    __amdgpu_buffer_rsrc_t row[3], rr, lab;
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pk_rsrc(const void *base, int64_t bytes) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    const int nb = __builtin_amdgcn_readfirstlane((int)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0, nb, 0x00020000);
}
constexpr int64_t kPkDualMaxN = 1ll << 28;  // 4 N bytes per row and every byte offset stay below 2^31
__device__ __forceinline__ void packed_issue_dual(const PkSrc &src, int64_t gg, unsigned lane_off, unsigned own_off, PkRows &o) {
    // byte offset of the lower block's four points of this column inside a row of dwords
    const unsigned oL = gg >= 0 ? (unsigned)gg * 1024u + lane_off : 0xfffffff0u;
    const unsigned oU = gg >= 0 ? oL + 512u : 0xfffffff0u;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        o.vL[p] = __builtin_amdgcn_raw_buffer_load_b128(src.row[p], oL, 0, 0);
        o.vU[p] = __builtin_amdgcn_raw_buffer_load_b128(src.row[p], oU, 0, 0);
    }
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t a = __builtin_amdgcn_raw_buffer_load_b64(src.rr, oL >> 1, 0, 0);
    const u32x2_t b = __builtin_amdgcn_raw_buffer_load_b64(src.rr, gg >= 0 ? oU >> 1 : 0xfffffff0u, 0, 0);
    o.rL[0] = a.x;
    o.rL[1] = a.y;
    o.rU[0] = b.x;
    o.rU[1] = b.y;
    o.lp = __builtin_amdgcn_raw_buffer_load_b32(src.lab, gg >= 0 ? (oL >> 2) + own_off : 0xfffffff0u, 0, 0);
}

// ---- the per-launch tables of packed_assign_body, as functions of a cluster's centroid (c[0..5], |c|^2 as stage_centroids
// sums it): the label-table row and the matrix operand of lane (col, half).  Made either inside packed_assign_body or --
// chained kernel -- by otherwise idle wavefronts beside the update's reductions (packed_tables_side).
__device__ __forceinline__ void pk_table_row(const float (&c)[6], float bn, const float *hdr, float s, float m_up, float *row) {
    constexpr float kUp = 1.001953125f;
    const float s2 = s * s;
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const float ct = c[i] - hdr[i];
        qq = fmaf(ct, ct, qq);
        row[i] = 2.0f * s * ct;
    }
    // (v_sqrt_f32, 1 ulp: both are upper bounds with a 1e-3 margin)
    const float Q = __builtin_amdgcn_sqrtf(qq) * s * 1.001f + 1e-30f, M = m_up + __builtin_amdgcn_sqrtf(bn) * s * 1.001f;
    row[6] = -(qq * s2);  // (negated: the chain starts from it)
    // th(R) = R^2 k1 + R thr_r + thr_1:  epsR + Ew_l + E1_l  (header comment), coefficients rounded up -- twice: the
    // second (1 + 2^-9), which covers the roundings of th's own evaluation, used to be a multiplication per point
    row[7] = (fmaf(9.86e-4f, Q, 9.7e-7f * M) * kUp + 1e-30f) * kUp;
    row[8] = ((fmaf(9.6e-7f * Q, Q, 2.4e-7f * Q) + fmaf(4.85e-7f * M, M, 1e-12f)) * kUp) * kUp;
}
// th's two cluster-independent coefficients with the same factor inside: 2^-22 |y| (the chain's rounding) and 7.4e-6 R^2
constexpr float kThY = 2.384185791015625e-7f * 1.001953125f, kThR2 = 7.4e-6f * 1.001953125f;

// A operand of a lane's cluster (layout as in filter_assign_body): lower half-wave lanes carry k-slots 0..7 =
// {hi(2 q)_0..5, -|q|^2 + const as hi, lo * 2^10}, upper half-wave lanes k-slots 8..15 = {lo(2 q)_0..5, slope, 0}
__device__ __forceinline__ u32x4 pk_a_operand(const float (&c)[6], float bn, bool valid, const float *hdr, float s, float m_up,
                                              int half) {
    constexpr float kUp = 1.001953125f, kTiny = 1.1920928955078125e-7f;  // (1 + 2^-9) v + 2^-23 survives the rtz to f16
    const float s2 = s * s;
    unsigned ch[3] = {0u, 0u, 0u}, cl[3] = {0u, 0u, 0u};
    float nb = -60000.0f;
    unsigned ebd = 0u;
    if (valid) {
        float ct[6], qq = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            ct[i] = c[i] - hdr[i];
            qq = fmaf(ct[i], ct[i], qq);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) split_f16(ct[2 * p], ct[2 * p + 1], 2.0f * s, ch[p], cl[p]);
        const float Q = sqrtf(qq) * s * 1.001f + 1e-30f, M = m_up + sqrtf(bn) * s * 1.001f;
        // u_j - epsR(R) = t'_j + R ebd_r + ebd_1
        const float ebd_r = fmaf(9.95e-4f, Q, fmaf(9.7e-7f, M, 4.8e-7f));
        // (+ 2e-10: the 2^-34 of E2 and, for a cluster so close to mu that -|q|^2 + const is positive, what the
        // round-toward-zero hi / lo pair below can fall short of it: < 2^-24 / 1024 = 5.8e-11)
        const float ebd_1 = fmaf(5.4e-6f * Q, Q, 4.8e-7f * Q) + fmaf(4.85e-7f * M, M, 2e-10f);
        nb = fmaf(-qq, s2, ebd_1 * kUp);
        nb = fmaf(fabsf(nb), 3.814697265625e-6f, nb) + 1e-12f;  // + 2^-18 |nb|: the hi / lo pair below never rounds it down
        ebd = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(fmaf(ebd_r, kUp, kTiny), 0.f));
    }
    const auto nh = __builtin_amdgcn_cvt_pkrtz(nb, 0.f);
    const unsigned bnd = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(nb, (nb - (float)nh[0]) * 1024.0f));
    return half == 0 ? u32x4{ch[0], ch[1], ch[2], bnd} : u32x4{cl[0], cl[1], cl[2], ebd};
}

// The tables above from the NEW centroids `cen` (d x K, LDS) while the update that made them is still reducing its error and
// flags: role 0 = label table (lanes < K), role 1 = matrix operand of the 64 (col, half) lanes -> tables[lane], role 2 =
// the exact rows for the drains (stage_centroids).  One wavefront per role; nothing written here overlaps the prologue's
// scratch (sC / sL lie behind the accumulator copies).  Speculative: if the update ends the fit or the launch falls back
// to the fp32 filter, the tables are simply not used.
__device__ __forceinline__ void packed_tables_side(int role, int lane, const float *cen, const float *hdr, int K, u32x4 *tables) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *sC = reinterpret_cast<float *>(smem_raw + sizeof(long long) * kPkAccCopies * kPkAccPitch);
    float *sL = sC + K * 8;
    const float s = hdr[6], m_up = hdr[7];
    if (role == 2) {
        if (lane < K) {
            float bn = 0.f;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float v = cen[i * K + lane];
                sC[lane * 8 + i] = v;
                bn = bn + v * v;  // kmeans.py:74 |b|^2: sequential sum of rounded squares (stage_centroids)
            }
            sC[lane * 8 + 6] = bn;
        }
        return;
    }
    const int col = lane & 31, half = lane >> 5;
    const int j = role == 0 ? lane : 2 * (4 * (col >> 3) + (col & 3)) + ((col >> 2) & 1);
    float c[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float bn = 0.f;
    if (j < K) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            c[i] = cen[i * K + j];
            bn = bn + c[i] * c[i];
        }
    }
    if (role == 0) {
        if (j < K) pk_table_row(c, bn, hdr, s, m_up, sL + j * kPkRow);
    } else {
        tables[lane] = pk_a_operand(c, bn, j < K, hdr, s, m_up, half);
    }
}

// (FIRST: the instantiation a fit's first launch takes -- the only one that can meet iteration 0, i.e. the exact scan that
// also writes the packed copy; the launches after it do not carry that code)
template <int NREGS, bool FIRST>
__device__ __forceinline__ void packed_assign_body(const LloydPacked pk, const float *hdr,
                                                   const float *__restrict__ X, int64_t N, int K,
                                                   const et_kmeans_state *state, const float *cen,
                                                   uint8_t *__restrict__ labels, long long *__restrict__ lanes,
                                                   int copy_mask, int range_bad = -1, const u32x4 *tables = nullptr) {
    const unsigned tx = thread_x();  // (opaque per call: see thread_x)
    constexpr int d = 6;
    const int n_thr = (int)blockDim.x, n_wav = n_thr >> 6;
    // (hdr: the caller's copy of *pk.hdr in LDS -- mu[6], s, mu_norm, ok --, requested together with the kernel's other
    // prologue loads: read here, it would be one more dependent round trip to memory in every launch)
    // (everything the decisions below read from LDS is requested at once: read where it is used -- behind one branch after
    // the other -- it was five dependent round trips, ~500 cycles of every launch's prologue)
    const int64_t st_iter = state->iter, st_fast_ok = state->fast_ok;
    const int frac = (int)state->frac;
    const float s = hdr[6], m_up = hdr[7];
    const unsigned pk_ok = __float_as_uint(hdr[8]);
    if (FIRST && st_iter <= 0 && pk.fused) {
        // the fit's first launch: the exact scan of every point -- which also writes the packed copy, from the rows it reads
        // anyway (the scan is bound by its arithmetic, ~130 us at 1e7 points, and has the memory side to spare: a pass of
        // its own over X, kmeans_pack_kernel, costs 175-190 us)
        PackOut po;
        po.xh = const_cast<unsigned *>(pk.xh);
        po.rr = const_cast<unsigned short *>(pk.rr);
        po.xa = const_cast<float4 *>(pk.xa);
        if (!packed_header(X, N, state, const_cast<PackedHeader *>(pk.hdr), po.mu, po.s)) po.xh = nullptr;
        assign_body_valu<6, 4>(X, N, d, K, state, cen, nullptr, labels, nullptr, lanes, copy_mask, po);
        return;
    }
    bool fallback = st_iter <= 0 || !st_fast_ok || pk_ok == 0u;
    fallback = fallback || N > kPkDualMaxN || (N & 3) != 0;  // (the rows are requested through 32-bit buffer offsets, 16 bytes at a time)
    if (!fallback) {  // every |s (c - mu)| inside the packed range?  (cen: d x K floats in LDS, the same in every workgroup)
        if (range_bad >= 0) {  // (the caller's update has looked already: uniform over the workgroup)
            fallback = range_bad != 0;
        } else {
            int bad = 0;
            for (int e = tx; e < d * K; e += n_thr) bad |= !(fabsf((cen[e] - hdr[e / K]) * s) < 31.0f);
            fallback = __syncthreads_or(bad) != 0;
        }
    }
    if (fallback) {
        filter_assign_body<NREGS, false>(X, N, K, state, cen, labels, nullptr, lanes, copy_mask);
        return;
    }
    const int plen = d * K + K + 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    long long *sAcc = reinterpret_cast<long long *>(smem_raw);                                 // kPkAccCopies x kPkAccPitch
    float *sC = reinterpret_cast<float *>(smem_raw + sizeof(long long) * kPkAccCopies * kPkAccPitch);  // K * 8: exact rows (drain)
    float *sL = sC + K * 8;                                                                    // K * kPkRow: label table
    const int lane = tx & 63, wave = tx >> 6, half = lane >> 5, col = lane & 31;
    unsigned *queue = reinterpret_cast<unsigned *>(sL + K * kPkRow) + wave * kPkQueue;
    u32x4 a1;
    __shared__ int sNext;
    if (tables) {
        // the caller's update made the tables beside its reductions (packed_tables_side): sC, sL and the operand are there
        if (tx == 0) sNext = n_wav;
        for (int i = tx; i < kPkAccCopies * kPkAccPitch; i += n_thr) sAcc[i] = 0;  // (the prologue scratch inside it is dead)
        a1 = tables[lane];
    } else {
        stage_centroids(cen, d, K, sC);
        if (tx == 0) sNext = n_wav;
        __syncthreads();  // (`cen` -- the chained kernel's prologue scratch -- lies inside the accumulator copies: cleared only now)
        for (int i = tx; i < kPkAccCopies * kPkAccPitch; i += n_thr) sAcc[i] = 0;
        // per cluster: the centred, scaled row for the old-label chain and the two threshold coefficients
        for (int j = tx; j < K; j += n_thr) {
            float c[d];
#pragma unroll
            for (int i = 0; i < d; ++i) c[i] = sC[j * 8 + i];
            pk_table_row(c, sC[j * 8 + 6], hdr, s, m_up, sL + j * kPkRow);
        }
        {
            const int j = 2 * (4 * (col >> 3) + (col & 3)) + ((col >> 2) & 1);
            float c[d] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            float bn = 0.f;
            if (j < K) {
#pragma unroll
                for (int i = 0; i < d; ++i) c[i] = sC[j * 8 + i];
                bn = sC[j * 8 + 6];
            }
            a1 = pk_a_operand(c, bn, j < K, hdr, s, m_up, half);
        }
    }
    const f16x8 A1 = __builtin_bit_cast(f16x8, a1);
    __syncthreads();  // sL complete
    KM_PSTAMP(5);
    const float4 *l4 = reinterpret_cast<const float4 *>(sL);

    int qn = 0;  // wave-uniform number of queued points
    const int64_t n_groups = (N + 255) / 256;
    const bool dynamic = n_groups > (int64_t)gridDim.x * n_wav;
    bool first = true;
    auto take = [&]() -> int64_t {  // this wavefront's next pass, or -1 (wave-uniform); see filter_assign_body
        int64_t g;
        if (first) {
            first = false;
            g = (int64_t)blockIdx.x * n_wav + wave;
        } else if (dynamic) {
            int i = 0;
            if (lane == 0) i = atomicAdd(&sNext, 1);
            i = __builtin_amdgcn_readfirstlane(i);
            g = (int64_t)blockIdx.x * n_wav + (i % n_wav) + (int64_t)(i / n_wav) * gridDim.x * n_wav;
        } else {
            g = n_groups;
        }
        return g < n_groups ? g : -1;
    };
    PkSrc src;
#pragma unroll
    for (int p = 0; p < 3; ++p) src.row[p] = pk_rsrc(pk.xh + (int64_t)p * N, 4 * N);
    src.rr = pk_rsrc(pk.rr, 2 * N);
    src.lab = pk_rsrc(labels, N);
    const unsigned lane_off = 16u * (unsigned)col, own_off = 128u * (unsigned)half;
    const unsigned ones = 0x14003c00u;  // {1, 2^-10}: partners of {hi, lo * 2^10} of -|q|^2 + const
    // one pass on the rows in `cu` (pass index gc); the other register set is in flight meanwhile
    auto process = [&](const PkRows &cu, int64_t gc) __attribute__((always_inline)) {
        const int64_t n = gc * 256 + 128 * half + 4 * col;
        const bool valid = n < N;
        const unsigned old_packed = cu.lp;
        unsigned undecided = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned w[3];
            u32x4 bLo, bUp;
            const unsigned rpL = cu.rL[q >> 1], rpU = cu.rU[q >> 1];
            // low half: the point's R (the high half meets a zero of the A operand)
            const unsigned r16L = (q & 1) ? (rpL >> 16) : rpL, r16U = (q & 1) ? (rpU >> 16) : rpU;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                bLo[p] = cu.vL[p][q];
                bUp[p] = cu.vU[p][q];
                w[p] = half ? bUp[p] : bLo[p];
            }
            bLo[3] = half ? r16L : ones;
            bUp[3] = half ? r16U : ones;
            const unsigned r16 = half ? r16U : r16L;
            // certified lower bound of the old label's value: fp32 chain on the f16 coordinates
            const int ol = (int)((old_packed >> (8 * q)) & 0xffu);
            const float4 r0 = l4[3 * ol], r1 = l4[3 * ol + 1], r2 = l4[3 * ol + 2];
            float y = r1.z;  // -|q_l|^2 s^2
            y = fma_mix_lo(w[0], r0.x, y);
            y = fma_mix_hi(w[0], r0.y, y);
            y = fma_mix_lo(w[1], r0.z, y);
            y = fma_mix_hi(w[1], r0.w, y);
            y = fma_mix_lo(w[2], r1.x, y);
            y = fma_mix_hi(w[2], r1.y, y);
            const f16x8 BL = __builtin_bit_cast(f16x8, bLo), BU = __builtin_bit_cast(f16x8, bUp);
            f32x16 accL, accU;
#pragma unroll
            for (int r = 0; r < 16; ++r) accL[r] = accU[r] = 0.f;
            accL = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, BL, accL, 0, 0, 0);
            accU = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, BU, accU, 0, 0, 0);
            float bL, sL_, bU, sU;
            top2<NREGS>(accL, bL, sL_);
            top2<NREGS>(accU, bU, sU);
            const auto rb = __builtin_amdgcn_permlane32_swap(__float_as_uint(bL), __float_as_uint(bU), false, false);
            const auto rq = __builtin_amdgcn_permlane32_swap(__float_as_uint(sL_), __float_as_uint(sU), false, false);
            const float b0 = __uint_as_float(rb[0]), b1 = __uint_as_float(rb[1]);
            const float s0 = __uint_as_float(rq[0]), s1 = __uint_as_float(rq[1]);
            const float second = vmed3(b0, b1, vmax(s0, s1));  // second largest upper bound
            // th(R) (1 + 2^-9): the factor is inside the coefficients (pk_table_row: r1.w, r2.x; kThY, kThR2 here)
            const float R = (float)__builtin_bit_cast(_Float16, (unsigned short)(r16 & 0xffffu));
            const float th = fmaf(fabsf(y), kThY, fmaf(R, fmaf(R, kThR2, r1.w), r2.x));
            const bool keep = y - second > th;
            undecided |= (valid && !keep) ? (1u << q) : 0u;
        }
        if (__ballot(undecided != 0u)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool push = (undecided >> q) & 1u;
                const unsigned long long m = __ballot(push);
                if (push) {
                    unsigned *e = queue + qn + __popcll(m & ((1ull << lane) - 1ull));
                    e[0] = (unsigned)(n + q);
                    e[kFilterSlots] = (old_packed >> (8 * q)) & 0xffu;
                }
                qn += __popcll(m);
#ifdef ET_FILTER_DEBUG
                if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[d * K + K + 1]), (unsigned long long)__popcll(m));
#endif
                if (qn >= 64) {
                    qn -= 64;
                    packed_drain(queue + qn, 64, K, sC, pk.xa, labels, sAcc, frac, lane);
                }
            }
        }
    };
    // the loop, unrolled by two over the register sets ra / rb: a set is requested one pass ahead and never copied
    PkRows ra, rb;
#ifdef ET_EXP_WAITSTAMP  // (passes and their cycles only: the exposed wait and the drains are no longer separable)
    unsigned long long ws_wait = 0, ws_pass = 0, ws_drain = 0, ws_n = 0, ws_nd = 0;
#define KM_WS_PASS(call)                                             \
    do {                                                             \
        const unsigned long long t0_ = __builtin_amdgcn_s_memtime(); \
        call;                                                        \
        ws_pass += __builtin_amdgcn_s_memtime() - t0_;               \
        ++ws_n;                                                      \
    } while (0)
#else
#define KM_WS_PASS(call) call
#endif
    int64_t g = take();
    packed_issue_dual(src, g, lane_off, own_off, ra);
    while (g >= 0) {
        const int64_t g2 = take();
        packed_issue_dual(src, g2, lane_off, own_off, rb);
        KM_WS_PASS(process(ra, g));
        if (g2 < 0) break;
        g = take();
        packed_issue_dual(src, g, lane_off, own_off, ra);
        KM_WS_PASS(process(rb, g2));
    }
#undef KM_WS_PASS
#ifdef ET_EXP_WAITSTAMP
    if (lane == 0) {  // (into LDS: six device atomics per wavefront here made the build's launches 4-5x slower)
        atomicAdd(&s_ws_acc[0], ws_n);
        atomicAdd(&s_ws_acc[1], ws_wait);
        atomicAdd(&s_ws_acc[2], ws_pass);
        atomicAdd(&s_ws_acc[3], ws_drain);
        atomicAdd(&s_ws_acc[4], ws_nd);
        atomicAdd(&s_ws_acc[6], 1ull);
    }
#endif
    KM_PSTAMP(6);
    if (qn) packed_drain(queue, qn, K, sC, pk.xa, labels, sAcc, frac, lane);
    __syncthreads();
    KM_PSTAMP(7);
    for (int i = tx; i < plen; i += n_thr) {  // the copies -> copy 0
        long long v = sAcc[i];
#pragma unroll
        for (int c = 1; c < kPkAccCopies; ++c) v += sAcc[c * kPkAccPitch + i];
        sAcc[i] = v;
    }
    __syncthreads();
    emit_partials(sAcc, plen, n_thr, nullptr, lanes, copy_mask);
#ifdef ET_EXP_WAITSTAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    KM_PSTAMP(8);
    KM_PSTAMP_FLUSH();
}
