// et_reforder_fast_geometry.inl -- part of csrc/et_kmeans_reforder.hip (ONE translation unit: this file is #included there, in order, and is
// not compiled on its own): the fast form (d = 6, K <= 32, 1024 <= N < 2^29): the cascade tree as geometry, workspace layout, the permuted copy, exact packed arg-max.
// =====================================================================================================================
// The FAST form of the reference-order Lloyd iteration (d = 6, K <= 32, 1024 <= N < 2^29): one launch per iteration.
//
// ATen's cascade (kmeans.py:180-182) is a fixed tree over INDEX RANGES, so it parallelises without changing a single
// addition: with L = level step, lane k in 0..3 and lane-term r <-> point n = 4 r + k,
//   level 0   a "chain" = the L consecutive lane terms of one (chunk, lane): sequential adds into the chunk's per-cluster
//             accumulators -- one work item per (chain, coordinate), the K accumulators in LDS ([cluster][chain]: the
//             lanes of a wavefront never share a bank), L read-add-write steps;
//   level 1   a "group" = L consecutive chunks (4 L^2 points): per (lane, coordinate, cluster) the chunk results are
//             added in chunk order -- one workgroup owns a group, so this never leaves LDS;
//   level 2   a "block" = L consecutive groups: folded, in group order, by whichever workgroup of the block arrives last;
//   level 3 + the leftovers (partial block / group / chunk, the N mod 4 terms), the lane combination, the division by the
//             count, the error in ATen's inner-sum order and the stop flag: by the workgroup that arrives last of all.
// Everything that crosses workgroups inside a launch travels through device-scope stores / loads / atomics (served by the
// memory side: no cache fence), arrivals are one relaxed atomic after s_waitcnt + barrier (the idiom of
// kmeans_lloyd_persist_kernel).  The same workgroup first ASSIGNS its group's points (exact arg-max, kmeans.py:143-158,
// norms in ATen's orders), so an iteration reads the coordinates once from memory.
//
// Layout: the points of the full groups are kept in a permuted copy XT made once per fit (reforder_permute_kernel): per
// group and coordinate the 4 L^2 values as [tile][r / 4][chain][r % 4] (tile = 16 chunks = 64 chains), so that a lane's
// 16-byte load is four consecutive steps of its own chain and a wavefront's load is 1 KB contiguous; labels live in the
// same order (LT) and are un-permuted once, when the fit hands them out.  The points after the last full group (< 4 L^2 +
// 4 L + 4: the "tail") stay where they are and belong to one extra workgroup.
//
// Several problems (blockIdx.y) iterate in ONE loop and stop TOGETHER on the error summed over the whole batch in ATen's
// inner-sum order over the contiguous (l, d, K) tensor -- kmeans.py:228-240.
// =====================================================================================================================
namespace fast {

constexpr int kD = 6;
constexpr int kFThreads = 384;  // six wavefronts: one per coordinate in the level-0 phase
constexpr int kFMaxK = 32;
constexpr int kFMaxBatch = 64;
constexpr int kFMaxLp = 6;  // L <= 64 (N < 2^29)
constexpr int kUThreads = 256;  // reforder_update_kernel2
constexpr size_t kUMaxLds = 128 * 1024;

struct Geo {
    int64_t N;
    int lp;               // L = 1 << lp
    int64_t G;            // full level-1 groups
    int64_t tail0;        // first point of the tail = G * 4 L^2
    int64_t full_chunks;  // (N / 4) / L
    int n_blk, full_blk;  // level-2 blocks (a partial last one included) / complete ones
};
static Geo make_geo(int64_t N, int lp_forced = 0) {  // lp_forced: a shard takes the level step of the WHOLE array
    Geo g;
    g.N = N;
    g.lp = lp_forced ? lp_forced : level_power(N / 4);
    const int64_t L = (int64_t)1 << g.lp;
    g.full_chunks = N / 4 / L;
    g.G = g.full_chunks / L;
    g.tail0 = g.G * 4 * L * L;
    g.full_blk = (int)(g.G / L);
    g.n_blk = (int)((g.G + L - 1) / L);
    return g;
}

// LDS of the groups kernel: [level-0 accumulators (K rows + a dummy one per coordinate and tile) | a group's label words];
// the tail's label bytes (4 L^2 + 4 L + 16) alias the accumulators until level 0 clears them
__host__ __device__ inline size_t acc_region_bytes(int K, int L, int TR) {
    const size_t acc = sizeof(float) * (size_t)TR * kD * (K + 1) * 64, tail = ((size_t)(4 * L * L + 4 * L + 16) + 15) / 16 * 16;
    return acc > tail ? acc : tail;
}

// byte offsets inside one problem's block of the workspace.  S1 / S2 / T hold one float4 = the four lanes k of a (group |
// block | tail part, coordinate, cluster) entry.
struct Layout {
    size_t state, cen, arrive, cnt, S1, S2, T, Sin, XT, LT, tail, bytes;
};
static Layout make_layout(const Geo &g, int K) {
    Layout l;
    const size_t dk = (size_t)kD * K;
    size_t off = 0;
    l.state = off;
    off = up(off + sizeof(et_kmeans_state));
    l.cen = off;
    off = up(off + sizeof(float) * dk);
    l.arrive = off;
    off = up(off + sizeof(unsigned) * 4);
    l.cnt = off;  // per workgroup of the groups kernel: its points per cluster
    off = up(off + sizeof(unsigned) * (size_t)(g.G + 1) * kFMaxK);
    l.S1 = off;
    off = up(off + sizeof(float4) * (size_t)g.G * dk);
    l.S2 = off;
    off = up(off + sizeof(float4) * (size_t)(g.n_blk + 1) * (dk + kFMaxK / 4));  // a row: d K sums, then the block's counts
    l.T = off;
    off = up(off + sizeof(float4) * (2 * dk + 1));
    l.Sin = off;
    off = up(off + sizeof(double) * (size_t)(g.G + 1));
    l.XT = off;
    off = up(off + sizeof(float) * (size_t)g.tail0 * kD);
    l.LT = off;
    off = up(off + (size_t)g.tail0 + 4);
    l.tail = off;
    off = up(off + (size_t)(g.N - g.tail0) + 4);
    l.bytes = off;
    return l;
}
// in front of the problems' blocks: the batch-wide arrival counter and the batch's squared centroid differences
static size_t shared_bytes(int K, int64_t batch) { return up(256 + sizeof(float) * (size_t)batch * kD * K); }

struct Args {
    const float *X;     // problem 0's points (d, N); problem b: X + b * x_stride
    int64_t x_stride;
    unsigned char *ws;  // problem 0's block; problem b: ws + b * ws_stride
    int64_t ws_stride;
    unsigned *batch_arrive;
    float *sq_all;      // (batch, d K) squared centroid differences of this iteration
    Layout lay;
    Geo geo;
    int K, batch;
    float tol;
    float *trace;       // (batch, max_iter, 2) or nullptr
    int max_iter;
    int tiles_per_round;  // level-0 tiles in LDS at a time (1 or 2)
    unsigned long long *mail;  // host-visible progress word or nullptr
};

template <typename T>
__device__ __forceinline__ T *at(unsigned char *ws, size_t off) { return reinterpret_cast<T *>(ws + off); }
__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_agent(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16-byte device-scope (sc1: served by the memory side, write-through) accesses through buffer instructions
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *base, int64_t bytes) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    const int nb = __builtin_amdgcn_readfirstlane((int)(bytes > 0x7fffffffll ? 0x7fffffffll : bytes));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0, nb, 0x00020000);
}
constexpr int kAuxSc1 = 1 << 4;  // gfx940+ cache-policy immediate: bit 0 sc0, bit 1 nt, bit 4 sc1
__device__ __forceinline__ float4 ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, kAuxSc1);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float4 f) {
    const u32x4_t v = {__float_as_uint(f.x), __float_as_uint(f.y), __float_as_uint(f.z), __float_as_uint(f.w)};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, kAuxSc1);
}

// X (d, N) -> XT: one work item per (group, tile, r / 4, chain): four strided reads per coordinate, one 16-byte store
__global__ __launch_bounds__(kThreads) void reforder_permute_kernel(const float *__restrict__ X, int64_t x_stride,
                                                                    unsigned char *ws, int64_t ws_stride, size_t off_XT,
                                                                    Geo geo) {
    X += (int64_t)blockIdx.y * x_stride;
    float4 *XT4 = reinterpret_cast<float4 *>(ws + (int64_t)blockIdx.y * ws_stride + off_XT);
    const int lp = geo.lp;
    const int64_t L = (int64_t)1 << lp, L2 = L * L;
    const int64_t total = geo.G * L2;  // quads
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = w >> (2 * lp), qi = w & (L2 - 1);
        const int t = (int)(qi & 63);
        const int64_t qrb = qi >> 6;                 // tile * (L / 4) + rb
        const int64_t q = qrb / (L / 4), rb = qrb % (L / 4);
        const int64_t c = q * 16 + (t >> 2);         // chunk inside the group
        const int64_t n0 = g * 4 * L2 + 4 * (c * L + 4 * rb) + (t & 3);
#pragma unroll
        for (int i = 0; i < kD; ++i) {
            const float *x = X + (int64_t)i * geo.N + n0;
            XT4[(g * kD + i) * L2 + qi] = make_float4(x[0], x[4], x[8], x[12]);
        }
    }
}

// arg-max over the K centroid rows in LDS (row j = c[0..5], |c_j|^2, -) for NP points; NANS: torch.max's rule (a NaN beats
// everything, the first one stays), else plain `>` (no similarity can be NaN).  The next row is requested while this one
// is evaluated.
template <bool NANS, int NP>
__device__ __forceinline__ void points_best(const float (&x)[NP][kD], const float (&an)[NP], const float *sC, int K, int (&lb)[NP],
                                            float (&bv)[NP]) {
    const float4 *s4 = reinterpret_cast<const float4 *>(sC);
    float4 n0 = s4[0], n1 = s4[1];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        lb[p] = 0;
        bv[p] = 0.f;
    }
    for (int j = 0; j < K; ++j) {
        const float4 c0 = n0, c1 = n1;
        if (j + 1 < K) {
            n0 = s4[2 * j + 2];
            n1 = s4[2 * j + 3];
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float y = fmaf(x[p][0], c0.x, 0.f);  // kmeans.py:71
            y = fmaf(x[p][1], c0.y, y);
            y = fmaf(x[p][2], c0.z, y);
            y = fmaf(x[p][3], c0.w, y);
            y = fmaf(x[p][4], c1.x, y);
            y = fmaf(x[p][5], c1.y, y);
            y = y * 2.0f;   // :72
            y = y - an[p];  // :73
            y = y - c1.z;   // :74
            const bool take = NANS ? (j == 0 || gt_nanmax(y, bv[p])) : (j == 0 || y > bv[p]);
            bv[p] = take ? y : bv[p];
            lb[p] = take ? j : lb[p];
        }
    }
}

// points_best<false> for the four points of a quad as two packed pairs (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: the
// same IEEE operations, two points per instruction).  No similarity can be NaN or infinite here (the caller checked the
// magnitudes), so "the first row always wins" is `y > -inf`.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void quad_best(const float4 (&xv)[kD], const float *sC, int K, int (&lb)[4], float (&bv)[4]) {
    f32x2 xa[kD], xb[kD];
#pragma unroll
    for (int i = 0; i < kD; ++i) {
        xa[i] = f32x2{xv[i].x, xv[i].y};
        xb[i] = f32x2{xv[i].z, xv[i].w};
    }
    f32x2 ana = xa[0] * xa[0], anb = xb[0] * xb[0];  // kmeans.py:73, a full block's column: rows in sequence (0 + s0 = s0)
#pragma unroll
    for (int i = 1; i < kD; ++i) {
        ana = ana + xa[i] * xa[i];
        anb = anb + xb[i] * xb[i];
    }
    int opaque = 0;  // (keeps the first rows' loads and their splats inside the caller's loop: hoisted, they cost 20 registers)
    asm volatile("" : "+v"(opaque));
    const float4 *s4 = reinterpret_cast<const float4 *>(sC) + opaque;
    float4 n0 = s4[0], n1 = s4[1];
    lb[0] = lb[1] = lb[2] = lb[3] = 0;
    bv[0] = bv[1] = bv[2] = bv[3] = -__builtin_inff();
    const f32x2 zero = {0.f, 0.f};
#pragma clang loop unroll(disable)
    for (int j = 0; j < K; ++j) {
        const float4 c0 = n0, c1 = n1;
        n0 = s4[2 * j + 2];  // (row K: the table has kFMaxK + 1 rows)
        n1 = s4[2 * j + 3];
        const float cc[kD] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y};
        f32x2 ya = zero, yb = zero;
#pragma unroll
        for (int i = 0; i < kD; ++i) {
            const f32x2 c = {cc[i], cc[i]};
            ya = __builtin_elementwise_fma(xa[i], c, ya);  // kmeans.py:71
            yb = __builtin_elementwise_fma(xb[i], c, yb);
        }
        ya = ya * 2.0f;  // :72
        yb = yb * 2.0f;
        ya = ya - ana;   // :73
        yb = yb - anb;
        const f32x2 bn = {c1.z, c1.z};
        ya = ya - bn;    // :74
        yb = yb - bn;
        const float y[4] = {ya.x, ya.y, yb.x, yb.y};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const bool take = y[p] > bv[p];
            bv[p] = take ? y[p] : bv[p];
            lb[p] = take ? j : lb[p];
        }
    }
}

__device__ __forceinline__ double wave_sum_f64(double v) {  // fixed tree: the same bits for the same inputs
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = v + __shfl_xor(v, o);
    return v;
}

#ifdef ET_EXP_RFSTAMP  // measurement build (tools/archive/rfstamp.py): s_memrealtime at the phase boundaries of four workgroups
__device__ unsigned long long g_rf_stamps[4 * 16];
#define RF_STAMP(who, i)                                                                                           \
    do {                                                                                                           \
        if ((who) < 4 && threadIdx.x == 0 && blockIdx.y == 0) g_rf_stamps[(who) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
// slot `i` of row `who`: the latest time any workgroup passed here
#define RF_STAMP_MAX(who, i)                                                                               \
    do {                                                                                                   \
        if (threadIdx.x == 0 && blockIdx.y == 0) atomicMax(&g_rf_stamps[(who) * 16 + (i)], __builtin_amdgcn_s_memrealtime()); \
    } while (0)
// slot `i` of row 3 += ticks since *t (thread 0 of workgroup 0 only), *t = now
#define RF_ACC(i, t)                                                                  \
    do {                                                                              \
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) {                 \
            const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();         \
            g_rf_stamps[3 * 16 + 8 + (i)] += now_ - (t);                              \
            (t) = now_;                                                               \
        }                                                                             \
    } while (0)
#else
#define RF_ACC(i, t) \
    do {             \
    } while (0)
#define RF_STAMP_MAX(who, i) \
    do {                     \
    } while (0)
#define RF_STAMP(who, i) \
    do {                 \
    } while (0)
#endif

// Levels 0 and 1 of the cascade for the chunks 0 .. n_all-1 of one group (n_all <= L), TR tiles (of 16 chunks) at a time:
//   level 0  wavefront = coordinate, lane = chain (chunk, lane k); the chain's K (+ one dummy) accumulators are the LDS
//            words [row][chain]; a step = read, add, write of the row its label names;
//   level 1  work item (coordinate, cluster): adds the results of the chunks < n_l1 in chunk order (four lanes k side by
//            side in one 16-byte read, four reads in flight); the result of chunk n_l1 (if n_all > n_l1: the lane terms after the last full chunk) is
//            handed back untouched in acc0.
// load(tile, rb, lane, coordinate) -> the four values of steps 4 rb .. 4 rb + 3 of chain `lane` of `tile`; sLab: the same
// steps' labels, one word per (tile, rb, chain); a label = K routes a term that does not exist to the dummy row.
template <class Load>
__device__ __forceinline__ void cascade_levels(Load load, const unsigned *sLab, float *sAcc, int K, int L, int TR, int n_all,
                                               int n_l1, float4 &acc1, float4 &acc0) {
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int RB = L / 4, rows = K + 1, dk = kD * K;
    const int tiles = (n_all + 15) >> 4;
    const int ci = tid / K, cj = tid % K;
    // Accumulator word of (row, chain): column chain ^ (4 (row & 7)) of the row's 64 words.  Level 0 (lane = chain, row =
    // label) stays inside bank  lane mod 4 + a scrambled multiple of 4; level 1 (lane = (coordinate, cluster), a 16-byte
    // read of the four lanes k of chunk c) finds the rows of eight consecutive clusters in eight different bank groups --
    // without the swizzle every lane of a wavefront reads the same four banks.
    [[maybe_unused]] unsigned long long tacc = __builtin_amdgcn_s_memrealtime();
    for (int q0 = 0; q0 < tiles; q0 += TR) {
        const int tr = tiles - q0 < TR ? tiles - q0 : TR;
        for (int ql = 0; ql < tr; ++ql) {
            float *blk = sAcc + ((size_t)(ql * kD + wave) * rows) * 64;  // this wavefront's (tile, coordinate) block
            {
                float4 *z = reinterpret_cast<float4 *>(blk);
                for (int e = lane; e < rows * 16; e += 64) z[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            RF_ACC(0, tacc);
            const unsigned *lr = sLab + (q0 + ql) * RB * 64 + lane;
            // the chain's values, four 16-byte loads (= 16 steps) in flight at a time: with one load per four steps the loop ran
            // at the latency of its loads, not of its LDS updates (eight in flight cost the registers of a seventh wavefront)
            for (int rb0 = 0; rb0 < RB; rb0 += 4) {
                float4 xc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) xc[u] = load(q0 + ql, rb0 + u, lane, wave);
#ifdef ET_EXP_RFSTAMP
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                RF_ACC(1, tacc);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    // four steps: their accumulators are requested together and the additions chained in registers -- a
                    // later step whose label repeats an earlier one takes that step's result instead of the (stale) word
                    // it read, and writes in order, so the row ends with the same sequential sum as read-add-write per
                    // step, at one LDS round trip per four steps instead of four
                    const float4 xv = xc[u];
                    const unsigned l4 = lr[(rb0 + u) * 64];
                    const unsigned j0 = l4 & 255u, j1 = (l4 >> 8) & 255u, j2 = (l4 >> 16) & 255u, j3 = l4 >> 24;
                    float *p0 = blk + j0 * 64 + (lane ^ ((j0 & 7u) << 2)), *p1 = blk + j1 * 64 + (lane ^ ((j1 & 7u) << 2));
                    float *p2 = blk + j2 * 64 + (lane ^ ((j2 & 7u) << 2)), *p3 = blk + j3 * 64 + (lane ^ ((j3 & 7u) << 2));
                    const float r0 = *p0, r1 = *p1, r2 = *p2, r3 = *p3;
                    const float n0 = r0 + xv.x;
                    const float n1 = (j1 == j0 ? n0 : r1) + xv.y;
                    const float n2 = (j2 == j1 ? n1 : (j2 == j0 ? n0 : r2)) + xv.z;
                    const float n3 = (j3 == j2 ? n2 : (j3 == j1 ? n1 : (j3 == j0 ? n0 : r3))) + xv.w;
                    *p0 = n0;
                    *p1 = n1;
                    *p2 = n2;
                    *p3 = n3;
                }
#ifdef ET_EXP_RFSTAMP
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                RF_ACC(2, tacc);
            }
        }
        __syncthreads();
        RF_ACC(3, tacc);
        if (tid < dk) {
            for (int ql = 0; ql < tr; ++ql) {
                const float *row = sAcc + ((size_t)(ql * kD + ci) * rows + cj) * 64;
                const int sw = (cj & 7) << 2;
#pragma clang loop unroll(disable)
                for (int h = 0; h < 4; ++h) {  // four chunks' results requested together, added in chunk order
                    float4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4 *>(row + (((4 * h + u) << 2) ^ sw));
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int cg = (q0 + ql) * 16 + 4 * h + u;
                        if (cg < n_l1) {
                            acc1.x = acc1.x + v[u].x;
                            acc1.y = acc1.y + v[u].y;
                            acc1.z = acc1.z + v[u].z;
                            acc1.w = acc1.w + v[u].w;
                        } else if (cg == n_l1) {
                            acc0 = v[u];
                        }
                    }
                }
            }
        }
        RF_ACC(4, tacc);
        __syncthreads();
        RF_ACC(5, tacc);
    }
}
