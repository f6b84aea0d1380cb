// et_kmeans_filter.inl -- part of csrc/et_kmeans.hip (ONE translation unit: this file is #included there, in order, and is not
// compiled on its own): the matrix-core label filter on fp32 rows (filter_assign_body): iterations >= 1 of traced fits and of shards below the packed threshold.
// clang-format off: the fragment starts and ends at namespace scope of whatever the including file has open.
// ------------------------------------------------------------------------------------------
// Lloyd half-step for iterations >= 1: matrix-core FILTER + exact certification (d = 6, K <= 32).
//
// The exact arg-max above costs ~12 VALU slots per (point, cluster) pair and that, not the 24 B/point
// read, is what the kernel waits for (SQ counters: 941 VALU instructions per 256 points, VALU busy 85 %).
// From the second iteration on almost every point keeps its label, and PROVING that is much cheaper
// than recomputing it:
//
//  1. f16 MFMA (v_mfma_f32_32x32x16_f16, ~16x the fp32 rate) evaluates t'_j ~ G_j = 2 x.c_j - |c_j|^2 for
//     all clusters: x and 2c, scaled by a power of two sg so that every magnitude is below 32, are split
//     into f16 (hi, lo) pairs (round to nearest: 2^-22 relative, 2^-25 absolute on the denormal grid -- the bounds
//     below are derived with the looser 2^-20 / 2^-24 of a round-toward-zero split);
//     the four partial products per coordinate and the split -|c|^2 occupy 26 of the 32 k-slots of two
//     MFMAs, accumulation is fp32 (<= 28 additions, order unknown: 2^-18.2 of the sum of magnitudes).
//     In scaled units, r = sg ||x||, C_j = sg ||c_j||:
//         |t'_j - G_j|            <= E2 = 2^-17.5 (r + C_j)^2 + 2^-21.7 (r + C_j) + 2^-34
//         |(Y_j + |x|^2) - G_j|   <= E1 = 2^-21 (r + C_j)^2           (fp32 chain of kmeans.py:71-74)
//     so with eps_j = 2^-16 (r + C_j)^2 + 2^-20 (r + C_j) + 2^-32 (> 2 (E1 + E2)):  Y_j + |x|^2 <= t'_j + eps_j.
//     eps_j = eps(r) + r (2^-15 C_j) + (2^-16 C_j^2 + 2^-20 C_j): the cluster-dependent part is linear in
//     (r, 1) and occupies two more k-slots, i.e. the MFMA delivers the UPPER BOUNDS u_j = t'_j + eps_j - eps(r)
//     (r and the coefficients rounded up).  A far-away centroid has a large error but an even more
//     negative u_j, so outliers do not loosen the test for ordinary points.
//  2. per point only the SECOND largest u is needed (no index): top-2 with v_max3/v_med3, 1.4 VALU
//     slots per pair.
//  3. the similarity Y_l of the point's OLD label l is evaluated exactly (one fmaf chain, needed for
//     the inertia anyway).  If  w = Y_l + |x|^2  exceeds  second + eps(r)  (+ the rounding of w), then
//     every cluster whose u is not the largest loses to l strictly, and l itself cannot be among them
//     (w <= u_l + eps(r)): l owns the largest u and is the reference's arg-max, strictly, no tie.  The
//     label is unchanged, nothing is accumulated (the sums are incremental), Y_l goes into the inertia.
//  4. every other point (label may change, or too close to call) is pushed on a per-wavefront LDS queue
//     and later gets the full exact scan, 64 queued points at a time, one per lane: label, exact
//     deltas, inertia -- exactly what kmeans_assign_kernel computes.
//
// The filter can only say "unchanged" when that is provably what the reference computes, so labels,
// sums and inertia stay bit-identical; its cost is ~430 VALU instructions per 256 points.
//
// MFMA layout: rows = clusters (A, loop invariant), columns = points (B).  A wavefront takes 256
// points per pass, lane (half, col) owning points 4 col..4 col+3 of its 128-point half.  Both halves
// of a column must feed the SAME point, so the owner's packed f16 dwords are broadcast across the
// halves with v_permlane32_swap (one VALU op yields both "lower half's value" and "upper half's
// value"); two tiles (lower points, upper points) per component q, and one more swap brings each
// half-wave the two partial (max, second) pairs of its own points.
// ------------------------------------------------------------------------------------------
// per-wavefront queue of undecided points: 8 rows (x[0..5], point index, old label) of kFilterSlots entries;
// < 64 entries are carried over and one component q of a pass adds at most 64
constexpr int kFilterSlots = 128;
constexpr int kFilterQueue = 8 * kFilterSlots;  // 32-bit words per wavefront

// full exact scan of `cnt` (<= 64) queued points, one per lane
template <bool SIM>
__device__ __forceinline__ void filter_drain(const unsigned *q, int cnt, int K, const float *sC,
                                             uint8_t *__restrict__ labels, long long *sAcc, int frac, int sfrac, int lane,
                                             long long &sim_acc) {
    constexpr int d = 6;
    if (lane >= cnt) return;
    const int64_t n = (int64_t)q[6 * kFilterSlots + lane];
    float x[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = __uint_as_float(q[i * kFilterSlots + lane]);
    const int old = (int)q[7 * kFilterSlots + lane];
    int lb;
    float best;
    best_centroid6_drain(x, sC, K, lb, best);
    if (lb != old) {
        labels[n] = (uint8_t)lb;
        atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[d * K + lb]), 1ull);
        atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[d * K + old]), ~0ull);
#pragma unroll
        for (int i = 0; i < d; ++i) {
            const unsigned long long f = (unsigned long long)to_fixed(x[i], frac);
            atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[i * K + lb]), f);
            atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[i * K + old]), 0ull - f);
        }
    }
    if (SIM) sim_acc += to_fixed(best, sfrac);
}

#ifdef ET_EXP_WAITSTAMP  // development aid (tools/archive/waitstamp.py): where a pass of packed_assign_body spends its time, in
// shader cycles (s_memtime) summed over all wavefronts and launches: [0] passes, [1] cycles from a pass's start to the
// arrival of its own (prefetched) rows = the exposed load wait, [2] cycles of whole passes, [3] cycles inside queue drains,
// [4] drains, [5] cycles from kernel start to the first pass, [6] wavefronts
__device__ unsigned long long g_waitstamp[8];
// ... and where a LAUNCH of the chained kernel goes (thread 0 of every workgroup, cycles between consecutive stamps, summed
// over workgroups and launches): [0] workgroup-launches, [1] entry -> prologue loads arrived, [2] fold + barrier, [3] update,
// [4] barrier + publish, [5] tables, matrix operand, barrier, [6] the pass loop, [7] final drain + barrier,
// [8] copies -> one + barrier + emit
__device__ unsigned long long g_prostamp[16];
__shared__ unsigned long long s_ps_last, s_ps_acc[16], s_ws_acc[8];
#define KM_PSTAMP(i)                                                        \
    do {                                                                    \
        if (threadIdx.x == 0) {                                             \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();     \
            s_ps_acc[i] = (i) ? t_ - s_ps_last : 1ull;                      \
            if ((i) == 0)                                                   \
                for (int z_ = 0; z_ < 8; ++z_) s_ws_acc[z_] = 0ull;         \
            s_ps_last = t_;                                                 \
        }                                                                   \
    } while (0)
#define KM_PSTAMP_FLUSH()                                                               \
    do {                                                                                \
        __syncthreads();                                                                \
        if (threadIdx.x == 0) {                                                         \
            for (int i_ = 0; i_ < 9; ++i_) atomicAdd(&g_prostamp[i_], s_ps_acc[i_]);     \
            for (int i_ = 0; i_ < 7; ++i_) atomicAdd(&g_waitstamp[i_], s_ws_acc[i_]);    \
        }                                                                               \
    } while (0)
#else
#define KM_PSTAMP(i)
#define KM_PSTAMP_FLUSH()
#endif

#ifdef ET_PERSIST_STAMPS  // development aid (tools/archive/persist_stamps.py): per workgroup and iteration, 10 ns ticks
// kinds 0..5 (persistent kernel): top (own arrival done), go, folded, updated, body start, body end;
// kinds 6..9 (inside the filter body): operands staged, passes done (wavefront 0), queue drained, deltas emitted
constexpr int kStampIters = 104, kStampKinds = 10;
__device__ unsigned long long g_persist_stamps[256 * kStampIters * kStampKinds];
__device__ int g_stamp_it[256];
#define ET_STAMP(kind)                                                                                              \
    do {                                                                                                            \
        if (threadIdx.x == 0 && blockIdx.x < 256 && it < kStampIters) {                                             \
            g_stamp_it[blockIdx.x] = it;                                                                            \
            g_persist_stamps[((size_t)blockIdx.x * kStampIters + it) * kStampKinds + (kind)] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                                                           \
    } while (0)
#define ET_BSTAMP(kind)                                                                                             \
    do {                                                                                                            \
        if (threadIdx.x == 0 && blockIdx.x < 256 && g_stamp_it[blockIdx.x] < kStampIters)                           \
            g_persist_stamps[((size_t)blockIdx.x * kStampIters + g_stamp_it[blockIdx.x]) * kStampKinds + (kind)] =  \
                __builtin_amdgcn_s_memrealtime();                                                                   \
    } while (0)
#else
#define ET_STAMP(kind) do { } while (0)
#define ET_BSTAMP(kind) do { } while (0)
#endif

// issue the loads of pass `gg` (256 points: lane (half, col) owns points 4 col .. 4 col + 3 of its 128-point half)
__device__ __forceinline__ void pass_issue(const float *__restrict__ X, int64_t N, const uint8_t *__restrict__ labels, int64_t gg,
                                           int half, int col, float4 (&vn)[6], unsigned &lpn) {
    const int64_t n = gg * 256 + 128 * half + 4 * col;
    // lanes past the end (last pass only) read points 0..3 instead: finite data, results discarded through `valid`
    // (unconditional loads: no exec-masked branch and no zero fill of 25 registers in every pass)
    const int64_t nl = n < N ? n : 0;  // N % 4 == 0
#ifdef ET_EXP_NOLOAD  // measurement aid (tools/archive/ab_lloyd.sh): the assignment without its memory traffic
    const float f = (float)(nl & 1023) * 0.01f;
#pragma unroll
    for (int i = 0; i < 6; ++i) vn[i] = make_float4(f + i, f - i, f * 0.5f, 1.0f + i);
    lpn = 0x01010101u * (unsigned)(nl & 7);
#else
#ifdef ET_EXP_NT_EVERY  // measurement aid: every ET_EXP_NT_EVERY-th pass bypasses the caches (does the rest then stay in the MALL?)
    if ((gg / 12) % ET_EXP_NT_EVERY == ET_EXP_NT_EVERY - 1) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float4 *p = reinterpret_cast<const float4 *>(X + (int64_t)i * N + nl);
            vn[i] = make_float4(__builtin_nontemporal_load(&p->x), __builtin_nontemporal_load(&p->y),
                                __builtin_nontemporal_load(&p->z), __builtin_nontemporal_load(&p->w));
        }
        lpn = *reinterpret_cast<const unsigned *>(labels + nl);
        return;
    }
#endif
#pragma unroll
    for (int i = 0; i < 6; ++i) vn[i] = *reinterpret_cast<const float4 *>(X + (int64_t)i * N + nl);
    lpn = *reinterpret_cast<const unsigned *>(labels + nl);
#endif
}

// The same for a HANDFUL of queued points (what a wavefront of a small shard holds at the end of its one pass: ~1 % of
// 256 points): one point per LANE leaves 60 lanes idle for a serial walk through the K centroids (2.1 us -- a sixth of a
// small shard's whole Lloyd iteration, profiles/r03f_persist_stamps_7e4.txt).  Here a half-wave takes one point and its
// lane j the similarity to centroid j -- the reference's operations in the reference's order, so the same bits --, the
// arg-max is a butterfly over the 32 lanes with torch.max's rule (first maximum wins; the filter body only runs when no
// similarity can be NaN / Inf, fast_ok), lanes 0..5 of the half-wave add the coordinate deltas.  Two points per step.
constexpr int kSmallDrain = 8;
template <bool SIM>
__device__ __forceinline__ void filter_drain_small(const unsigned *q, int cnt, int K, const float *sC,
                                                   uint8_t *__restrict__ labels, long long *sAcc, int frac, int sfrac, int lane,
                                                   long long &sim_acc) {
    constexpr int d = 6;
    const int j = lane & 31, hw = lane >> 5;
    const float4 *s4 = reinterpret_cast<const float4 *>(sC);  // rows of 8 floats: c[0..5], |c|^2, -
    const int jr = j < K ? j : 0;
    const float4 c0 = s4[2 * jr], c1 = s4[2 * jr + 1];
    for (int p0 = 0; p0 < cnt; p0 += 2) {
        const int p = p0 + hw;
        const bool live = p < cnt;
        const int ps = live ? p : 0;
        float x[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) x[i] = __uint_as_float(q[i * kFilterSlots + ps]);
        float an = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) an = an + x[i] * x[i];  // kmeans.py:73 |a|^2
        float y = fmaf(x[0], c0.x, 0.f);                     // :71
        y = fmaf(x[1], c0.y, y);
        y = fmaf(x[2], c0.z, y);
        y = fmaf(x[3], c0.w, y);
        y = fmaf(x[4], c1.x, y);
        y = fmaf(x[5], c1.y, y);
        y = y * 2.0f;   // :72
        y = y - an;     // :73
        y = y - c1.z;   // :74
        int lb = j;
        if (j >= K) y = -__int_as_float(0x7f800000);  // no such centroid: loses to every finite similarity
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            const float oy = __shfl_xor(y, o);
            const int ol = __shfl_xor(lb, o);
            if (oy > y || (oy == y && ol < lb)) {  // first maximum wins (kmeans.py:156: torch.max)
                y = oy;
                lb = ol;
            }
        }
        if (live) {
            const int old = (int)q[7 * kFilterSlots + p];
            if (lb != old) {
                if (j == 0) {
                    labels[(int64_t)q[6 * kFilterSlots + p]] = (uint8_t)lb;
                    atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[d * K + lb]), 1ull);
                    atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[d * K + old]), ~0ull);
                }
                if (j < d) {
                    const float xi = j == 0 ? x[0] : (j == 1 ? x[1] : (j == 2 ? x[2] : (j == 3 ? x[3] : (j == 4 ? x[4] : x[5]))));
                    const unsigned long long f = (unsigned long long)to_fixed(xi, frac);
                    atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[j * K + lb]), f);
                    atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[j * K + old]), 0ull - f);
                }
            }
            if (SIM && j == 0) sim_acc += to_fixed(y, sfrac);
        }
    }
}

// SIM = false: the similarity sum (the inertia of THIS assignment, kmeans.py:234) is not accumulated -- a fit that
// does not record the per-iteration trace evaluates the inertia once, after its last assignment
// (kmeans_inertia_kernel); the labels and the cluster sums are the same either way.
template <int NREGS, bool SIM = true>
__device__ __forceinline__ void filter_assign_body(const float *__restrict__ X, int64_t N, int K,
                                                   const et_kmeans_state *state, const float *cen,
                                                   uint8_t *__restrict__ labels, long long *__restrict__ block_partials,
                                                   long long *__restrict__ lanes = nullptr,
                                                   int copy_mask = kAccLanes - 1) {
    const unsigned tx = thread_x();  // (opaque per call: see thread_x)
    constexpr int d = 6;
    // power-of-two scale: every |x| sg, |c| sg < 32, so that |2c x| sg^2 < 6 * 2^11 and |c|^2 sg^2 < 6 * 2^10 fit
    // f16 and stay far above the -60000 that pads the rows of clusters >= K
    const int e_max = exponent_above(fmax(state->max_abs_x, state->max_abs_c));
    // first iteration (no labels yet), possible NaN/Inf, or a scale whose square leaves the fp32 range:
    // the exact kernel decides
    if (state->iter <= 0 || !state->fast_ok || e_max < -40 || e_max > 60) {
        assign_body_valu<6, 4>(X, N, d, K, state, cen, nullptr, labels, block_partials, lanes, copy_mask);
        return;
    }
    const int plen = d * K + K + 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    long long *sAcc = reinterpret_cast<long long *>(smem_raw);                                 // plen
    float *sC = reinterpret_cast<float *>(smem_raw + sizeof(long long) * ((plen + 1) & ~1));  // K * 8
    const int lane = tx & 63, wave = tx >> 6, half = lane >> 5, col = lane & 31;
    unsigned *queue = reinterpret_cast<unsigned *>(sC + K * 8) + wave * kFilterQueue;
    const int frac = (int)state->frac, sfrac = (int)state->sim_frac;
    // threshold polynomial in rr, already multiplied by sg^2 (the MFMA works on scaled operands)
    constexpr float kUp = 1.001953125f, kTiny = 1.1920928955078125e-7f;  // (1 + 2^-9) v + 2^-23 survives the rtz to f16
    // Trace-less fits (SIM = false) certify with w' = 2 x.c_l - |c_l|^2 instead of the reference's Y_l + |x|^2 (no |x|^2
    // chain, no square root).  With u = 2^-24, a = 2 d - an, Y_l = fl(fl(a) - cc):  Y_l + an = (2 d - cc) + a d1 +
    // (fl(a) - cc) d2, |d1|, |d2| <= u, so |Y_l + an - (2 d - cc)| <= u (2 |a| + cc)(1 + u) <= 3 u (r + C)^2 (1 + 2^-18)
    // <= 2^-21 (r^2 + C^2) (1 + 2^-10); w' itself adds one rounding, u |w'|.  r is bounded by sqrt(6) max|x_i|.
    constexpr float kSqrt6Up = 2.4543f;       // sqrt(6) (1 + 2^-9)
    constexpr float kR2Slack = 1.57365e-5f;   // 2^-16 + 2^-21 (1 + 2^-10), rounded up: eps(r)'s r^2 term + the slack above
    constexpr float kCcSlack = 4.7731e-7f;    // 2^-21 (1 + 2^-10), rounded up
    const float sg = ldexpf(1.0f, 5 - e_max), sg2 = sg * sg;
    const float sgk = sg * kSqrt6Up;  // exact (sg is a power of two, 2^-55 ... 2^45)
    const int n_thr = (int)blockDim.x, n_wav = n_thr >> 6;  // 768 or 1024 threads (host's choice)
    for (int i = tx; i < plen; i += n_thr) sAcc[i] = 0;
    stage_centroids(cen, d, K, sC);
    // slot 7 of a centroid row: the |c|^2 part of the rounding slack of the trace-less certification (below)
    for (int j = tx; j < K; j += n_thr) sC[j * 8 + 7] = fmaf(sC[j * 8 + 6] * sg2, kCcSlack, 2.3283064365386963e-10f);
    __shared__ int sNext;  // next of this workgroup's passes: the wavefronts take them as they come (see the loop)
    if (tx == 0) sNext = (int)(blockDim.x >> 6);  // a wavefront's first pass is its own (static), the others are handed out
    __syncthreads();


    // A operands: this lane feeds accumulator row m = col, k-half = half.  Row m is read back by lanes
    // of half (m >> 2) & 1 in register 4 (m >> 3) + (m & 3); cluster j sits in register j >> 1 of half j & 1,
    // so both halves reduce over registers 0 .. ceil(K / 2) - 1 <= NREGS - 1 (rows of clusters >= K: -60000).
    u32x4 a1 = {0u, 0u, 0u, 0u}, a2 = {0u, 0u, 0u, 0u};
    {
        const int j = 2 * (4 * (col >> 3) + (col & 3)) + ((col >> 2) & 1);
        unsigned ch[3] = {0u, 0u, 0u}, cl[3] = {0u, 0u, 0u};
        float nb = -60000.0f;
        if (j < K) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
                split_f16(sC[j * 8 + 2 * p], sC[j * 8 + 2 * p + 1], 2.0f * sg, ch[p], cl[p]);
            nb = -sC[j * 8 + 6] * sg2;
        }
        // -|c|^2 = hi + lo; lo is carried as lo * 2^10 against a 2^-10 on the point side (finer f16 grid)
        const auto nh = __builtin_amdgcn_cvt_pkrtz(nb, 0.f);
        const unsigned bnd = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(nb, (nb - (float)nh[0]) * 1024.0f));
        // the cluster-dependent part of the error bound, eps_j - eps(r) = r (2^-15 C_j) + (2^-16 C_j^2 + 2^-20 C_j)
        // with C_j = sg ||c_j|| rounded up, rides along as two more k-slots against (r, 1)
        unsigned ebd = 0u;
        if (j < K) {
            const float cj = sqrtf(sC[j * 8 + 6]) * sg * 1.001f;
            ebd = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(fmaf(3.0517578125e-5f * cj, kUp, kTiny),
                                                                           fmaf(fmaf(cj, 1.52587890625e-5f, 9.5367431640625e-7f) * cj, kUp, kTiny)));
        }
        // k-slots: the lower half-wave's lanes meet the points' hi parts (and the {1, 2^-10} of -|c|^2), the upper
        // half-wave's lanes the lo parts (and (r, 1)); the first MFMA multiplies both by hi(2c), the second by lo(2c)
        a1 = u32x4{ch[0], ch[1], ch[2], half == 0 ? bnd : ebd};
        a2 = u32x4{cl[0], cl[1], cl[2], 0u};
    }
    const f16x8 A1 = __builtin_bit_cast(f16x8, a1), A2 = __builtin_bit_cast(f16x8, a2);
    ET_BSTAMP(6);
    const float4 *s4 = reinterpret_cast<const float4 *>(sC);

    // Inertia: trunc(Y 2^sim_frac) is an integer below 2^(62 - bits(n_total)); a lane may add 2^(bits - 9) of them
    // in fp64 without leaving the exactly representable integers (< 2^53), four fp64 ops per point instead of
    // the ~25 of the integer conversion.  Flushed into the 64-bit accumulator before that limit.
    long long sim_acc = 0;
    double dsum = 0.0;
    const double sim_scale = ldexp(1.0, sfrac);
    const int term_limit = 1 << min(max(bits_for(state->n_total) - 9, 2), 30);
    int terms = 0;
    int qn = 0;  // wave-uniform number of queued points
    const int64_t n_groups = (N + 255) / 256;
    // The workgroup's passes (the same set as with a fixed wavefront -> pass map) are handed out through an LDS counter:
    // a SIMD serves its oldest wavefront first, so with a fixed map the first wavefront of a SIMD finished its share at
    // 27 us of a 41 us assignment phase and the SIMD ran its tail with two, then one wavefront.  Sums are exact integers
    // and labels per point, so who processes a pass does not matter.
    // (shards with at most one pass per wavefront keep the fixed map: nothing to balance, and no counter round trip)
    const bool dynamic = n_groups > (int64_t)gridDim.x * n_wav;
    // (Spreading the passes of the last, partial round evenly over all workgroups -- 152.6 each instead of 156 for
    // workgroups 0..182 and 144 for the rest at N = 1e7 -- was measured and dropped: +0.9 us per launch.  A launch's
    // workgroups start over ~5.7 us in index order, so the ones with the extra round are the ones that start first.
    // Leaning into that -- run lengths of the last rounds falling linearly with the workgroup index, 0.04 ... 0.16 passes
    // per index -- lost as well: 50.7 / 50.6 / 51.5 / 52.2 against 50.2 us.)
    bool first = true;
    auto take = [&]() -> int64_t {  // this wavefront's next pass, or -1 (wave-uniform)
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
    // (Requesting a pass's coordinates one pass ahead, or a wavefront's first pass before the persistent kernel's grid
    // barrier -- 25 more VGPRs each, free at three wavefronts per SIMD -- was measured and dropped: 51.4 against 50.4 us
    // per chained launch at N = 1e7 and no change of the persistent iteration; with the matrix-core work AND the top-2
    // removed the launch still takes 49.5 us (tools/archive/ab_lloyd.sh, profiles/r03e_ab_lloyd.txt): the passes run at what the
    // memory side delivers for this access pattern, ~6 TB/s, and are neither latency nor issue bound.)
    float4 vn[6];
    unsigned lpn = 0u;
    int64_t g = take();
    if (g >= 0) pass_issue(X, N, labels, g, half, col, vn, lpn);
    while (g >= 0) {
        const int64_t n = g * 256 + 128 * half + 4 * col;
        const bool valid = n < N;
        float4 v[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = vn[i];
        const unsigned old_packed = lpn;
        unsigned undecided = 0u;  // bit q: point q of this lane goes to the queue
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float x[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) x[i] = q == 0 ? v[i].x : (q == 1 ? v[i].y : (q == 2 ? v[i].z : v[i].w));
            float an = 0.f, rs;
            if constexpr (SIM) {
#pragma unroll
                for (int i = 0; i < 6; ++i) an = an + x[i] * x[i];  // kmeans.py:73
                rs = fmaf(__builtin_amdgcn_sqrtf(an) * sg, kUp, kTiny);  // >= sg ||x||
            } else {
                const float m = vmax(max3_abs(x[0], x[1], x[2]), max3_abs(x[3], x[4], x[5]));
                rs = fmaf(m, sgk, kTiny);  // = fl(m sg kSqrt6Up + kTiny) >= sg ||x|| as well: ||x|| <= sqrt(6) max |x_i|
            }
            unsigned w[7];  // {xh01, xh23, xh45, xl01, xl23, xl45, (r, 1)}
#pragma unroll
            for (int p = 0; p < 3; ++p) split_f16(x[2 * p], x[2 * p + 1], sg, w[p], w[3 + p]);
            w[6] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(fmaf(rs, kUp, kTiny), 1.0f));
            // One v_permlane32_swap of (hi, lo) dwords yields the B dword of both tiles: r[0] = {lower lanes: own hi,
            // upper lanes: the lower partner's lo} feeds tile L (points of the lower half-wave), r[1] = {lower lanes:
            // the upper partner's hi, upper lanes: own lo} feeds tile U -- no copies, and the same four dwords serve
            // both MFMAs (the second one multiplies the fourth by zero).
            const unsigned ones = 0x14003c00u;  // {1, 2^-10}: partners of {hi, lo * 2^10} of -|c|^2
            u32x4 bLo, bUp;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const auto r = p < 3 ? __builtin_amdgcn_permlane32_swap(w[p], w[3 + p], false, false)
                                     : __builtin_amdgcn_permlane32_swap(ones, w[6], false, false);
                bLo[p] = r[0];
                bUp[p] = r[1];
            }
            const f16x8 BL = __builtin_bit_cast(f16x8, bLo), BU = __builtin_bit_cast(f16x8, bUp);
            f32x16 accL, accU;
#pragma unroll
            for (int r = 0; r < 16; ++r) accL[r] = accU[r] = 0.f;
            float bL, sL, bU, sU;
#ifdef ET_EXP_NOMFMA  // measurement aid: no matrix-core work and no top-2 (every point is "kept")
            bL = __uint_as_float(bLo[0]) * 1e-30f, sL = __uint_as_float(bLo[1]) * 1e-30f - 1e30f;
            bU = __uint_as_float(bUp[2]) * 1e-30f, sU = __uint_as_float(bUp[3]) * 1e-30f - 1e30f;
#else
            accL = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, BL, accL, 0, 0, 0);
            accU = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, BU, accU, 0, 0, 0);
            accL = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2, BL, accL, 0, 0, 0);
            accU = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2, BU, accU, 0, 0, 0);
            top2<NREGS>(accL, bL, sL);
            top2<NREGS>(accU, bU, sU);
#endif
            // lower half-wave: both partials of its own points (tile L); upper half-wave: those of tile U
            const auto rb = __builtin_amdgcn_permlane32_swap(__float_as_uint(bL), __float_as_uint(bU), false, false);
            const auto rq = __builtin_amdgcn_permlane32_swap(__float_as_uint(sL), __float_as_uint(sU), false, false);
            const float b0 = __uint_as_float(rb[0]), b1 = __uint_as_float(rb[1]);
            const float s0 = __uint_as_float(rq[0]), s1 = __uint_as_float(rq[1]);
            const float second = vmed3(b0, b1, vmax(s0, s1));  // second largest upper bound u_j
            // exact similarity to the old label's (updated) centroid, kmeans.py:71-74
            const int ol = (int)((old_packed >> (8 * q)) & 0xffu);
            const float4 r0 = s4[2 * ol], r1 = s4[2 * ol + 1];
            float y = 0.f;
            y = fmaf(x[0], r0.x, y);
            y = fmaf(x[1], r0.y, y);
            y = fmaf(x[2], r0.z, y);
            y = fmaf(x[3], r0.w, y);
            y = fmaf(x[4], r1.x, y);
            y = fmaf(x[5], r1.y, y);
            float wv, th;
            if constexpr (SIM) {
                y = y * 2.0f;
                y = y - an;
                y = y - r1.z;
                // keep <=> (Y_l + |x|^2) sg^2 exceeds every other cluster's upper bound: w - second > eps(r) + rounding of w
                wv = (y + an) * sg2;
                th = fmaf(fabsf(wv), 2.384185791015625e-7f, fmaf(rs, fmaf(rs, 1.52587890625e-5f, 9.5367431640625e-7f), 2.3283064365386963e-10f));
            } else {
                // the same test on a certified lower bound of Y_l + |x|^2 (see kR2Slack above; r1.w = the |c_l|^2 slack)
                wv = fmaf(y, 2.0f, -r1.z) * sg2;
                th = fmaf(fabsf(wv), 1.1920928955078125e-7f, fmaf(rs, fmaf(rs, kR2Slack, 9.5367431640625e-7f), r1.w));
            }
            const bool keep = wv - second > th;
            if (SIM) {
                const double term = trunc((double)y * sim_scale);
                dsum += (valid && keep) ? term : 0.0;
            }
            undecided |= (valid && !keep) ? (1u << q) : 0u;
        }
        if (__ballot(undecided != 0u)) {  // rare once Lloyd settles: queue the points that need the full scan
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool push = (undecided >> q) & 1u;
                const unsigned long long m = __ballot(push);
                if (push) {  // the coordinates travel with the entry: no second trip to HBM for them
                    unsigned *e = queue + qn + __popcll(m & ((1ull << lane) - 1ull));
#pragma unroll
                    for (int i = 0; i < 6; ++i)
                        e[i * kFilterSlots] = __float_as_uint(q == 0 ? v[i].x : (q == 1 ? v[i].y : (q == 2 ? v[i].z : v[i].w)));
                    e[6 * kFilterSlots] = (unsigned)(n + q);
                    e[7 * kFilterSlots] = (old_packed >> (8 * q)) & 0xffu;
                }
                qn += __popcll(m);
#ifdef ET_FILTER_DEBUG
                if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[d * K + K + 1]), (unsigned long long)__popcll(m));
#endif
                if (qn >= 64) {
                    qn -= 64;
                    filter_drain<SIM>(queue + qn, 64, K, sC, labels, sAcc, frac, sfrac, lane, sim_acc);
                }
            }
        }
        if (SIM) {
            terms += 4;
            if (terms + 4 > term_limit) {
                sim_acc += (long long)dsum;
                dsum = 0.0;
                terms = 0;
            }
        }
        g = take();
        if (g >= 0) pass_issue(X, N, labels, g, half, col, vn, lpn);
    }
    sim_acc += (long long)dsum;
    ET_BSTAMP(7);
    if (qn > kSmallDrain) filter_drain<SIM>(queue, qn, K, sC, labels, sAcc, frac, sfrac, lane, sim_acc);
    else if (qn) filter_drain_small<SIM>(queue, qn, K, sC, labels, sAcc, frac, sfrac, lane, sim_acc);
    ET_BSTAMP(8);
    if (SIM) {
        for (int o = 32; o > 0; o >>= 1) sim_acc += __shfl_xor(sim_acc, o);
        if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[d * K + K]), (unsigned long long)sim_acc);
    }
    __syncthreads();
    emit_partials(sAcc, plen, n_thr, block_partials, lanes, copy_mask);
    ET_BSTAMP(9);
}

// ------------------------------------------------------------------------------------------
// Trace-less Lloyd iterations on a PACKED copy of the points (d = 6, K <= 32, big shards).
//
