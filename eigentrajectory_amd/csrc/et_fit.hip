// et_fit.hip -- descriptor fit: Gram matrices of the normalised trajectory block and
// their eigendecomposition (replaces torch.linalg.svd at EigenTrajectory/descriptor.py:109-114:
// the left singular vectors of the (2T x N) matrix M are the eigenvectors of M M^T and
// sigma = sqrt(lambda); the (N x k) right factor the reference also builds is never used,
// descriptor.py:134-135).
//
// Numerics: products of two fp32 values are exact in fp64, so the Gram matrix is
// accumulated in fp64 (an fp32 Gram costs ~1e-4 in U because forming M M^T squares the
// condition number).  Partial sums are combined in a fixed order
// (workgroup partials -> one reduction kernel), so a given (N, shard) is reproducible.
#include "et_common.h"

namespace et {

constexpr int kFitThreads = 256;

template <int TO, int TP>
struct GramLayout {
    static constexpr int DO = 2 * TO, DP = 2 * TP, D = DO + DP;
    // ten v_mfma_f64_4x4x4_4b per four rows: an instruction is four independent 4 x 4 x 4 products ("slots"), slot s working
    // on FOUR blocks of four features (gram4_block) and instruction t on the pair (qa <= qb) of them -- every slot computes
    // the upper triangle of its 16 x 16 sub-Gram.  Slot 0: the observation's blocks 0..3; slots 1..3: the prediction's
    // {0,1,2,3}, {4,5,0,1}, {2,3,4,5} -- three 4-subsets that cover every pair of its six blocks.  40 block products per four
    // rows against the 48 of three 16 x 16 tiles, at the same rate per product (tools/exp/mfma_f64_rate.hip: 16 against
    // 64 cycles), the same four operand conversions per lane.
    static constexpr int kPairs = 10;
    static constexpr int kSums = kPairs * 64;  // [pair t][lane]: lane = j + 4 slot + 16 i (i: row of the A block, j: row of the B block)
    static constexpr int kPartial = kSums + 1;  // doubles per partial (+ row count)
};
// unified feature block (0..3 observation, 4..9 prediction) at position q of slot s
__host__ __device__ constexpr int gram4_block(int s, int q) { return s == 0 ? q : 4 + (s == 1 ? q : (s == 2 ? (q < 2 ? 4 + q : q - 2) : 2 + q)); }
__host__ __device__ constexpr int gram4_pair(int qa, int qb) { return qa * 4 - qa * (qa - 1) / 2 + (qb - qa); }  // qa <= qb

// Gram kernel, (T_obs, T_pred) = (8, 12).  A wavefront takes 64 trajectories per pass, in its own 11 KB LDS slice:
//  1. coalesced float4 loads (prefetched one pass ahead into registers) -> one LDS row per trajectory, [obs 16 | pred 24]
//     + 16 B (pitch 11 float4: odd)
//  2. lane = trajectory: normaliser state, normalise the row IN PLACE (fp32; rows that do not belong
//     to descriptor `which` become zeros)
//  3. fp64 matrix instructions: per group of 4 trajectories ten v_mfma_f64_4x4x4_4b_f64 add the outer products of
//     4-feature blocks to ten accumulators (GramLayout; operand lane = i + 4 slot + 16 k: feature i of the slot's block, of
//     trajectory k; result lane = j + 4 slot + 16 i -- found by experiment, tools/exp/mfma_f64_4x4x4_layout.hip).  Products
//     of fp32 values are exact in fp64 and every entry is one k-ordered fma chain.  The loop over the wavefront's 16 groups
//     is unrolled: the LDS operand reads use immediate offsets.
//  4. waves are summed through LDS in a fixed order; every workgroup writes one partial.
// On this chip fp64 matrix instructions and vector instructions do not overlap, not even from different wavefronts of a
// SIMD (tools/exp/mfma_f64_rate.hip: 2 x 16x16x4 beside 32 v_fma_f32 of a second wavefront take the SUM of their times;
// the fp64 matrix rate equals the fp64 vector rate), so the kernel is built for the fewest vector instructions per row --
// lane = trajectory normalisation, 5.7 VALU instructions per row, four v_cvt_f64_f32 (6 ns each, as much as 256 fp64
// multiply-adds) per lane and group -- and, the slices being wavefront-private, there is no workgroup barrier inside the
// loop: the three resident wavefronts of a SIMD drift apart and fill each other's waits.
// (Round 2 A/B at N = 1e7, gram + reduce + finish: 256-row workgroup tiles with three barriers per pass 496-506 us;
// operands loaded straight into the MFMA layout and normalised there with DPP, no LDS staging, 6 wavefronts per SIMD:
// 484-490 us, 11 VALU instructions per row; wavefront slices + three v_mfma_f64_16x16x4 per group -- 48 block products for
// the 31 distinct ones, rounds 2-6: 457-464 -> 412-423 us; ten 4x4x4_4b per group -- 40 block products: 408-416 us, the
// fit stage inside the bench step 0.497-0.501 against 0.511-0.533 ms, profiles/r06p_gram4_ab.txt.)
constexpr int kWaveGramThreads = 256;

// normalize_point (et_common.h) on two points at once with the packed fp32 instructions (v_pk_add / v_pk_mul_f32: two
// results per lane and issue slot): (tx c, ty c) + (ty s, tx (-s)), then * sca -- the same products and the same single
// additions as normalize_point, hence the same bits (sca = 1 for the static descriptor: x * 1 is x).  Five instructions
// per point instead of ten.
#ifndef ET_EXP_GRAM_NT
#define ET_EXP_GRAM_NT 0
#endif
#ifndef ET_EXP_GRAM
#define ET_EXP_GRAM 0
#endif
typedef float f32x4_ld __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_rows(const float4 *p) {
#if ET_EXP_GRAM_NT
    const f32x4_ld v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_ld *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
typedef float f32x2_t __attribute__((ext_vector_type(2)));
struct NormPk {
    f32x2_t o, cc, sn, sc;
    __device__ __forceinline__ explicit NormPk(const RowNorm &p)
        : o{p.ox, p.oy}, cc{p.c, p.c}, sn{p.s, -p.s}, sc{p.mv ? p.sca : 1.0f, p.mv ? p.sca : 1.0f} {}
    __device__ __forceinline__ f32x2_t point(f32x2_t xy) const {
        const f32x2_t t = xy - o;
        return (t * cc + __builtin_shufflevector(t, t, 1, 0) * sn) * sc;
    }
    __device__ __forceinline__ float4 quad(float x0, float y0, float x1, float y1) const {
        const f32x2_t a = point((f32x2_t){x0, y0}), b = point((f32x2_t){x1, y1});
        return make_float4(a.x, a.y, b.x, b.y);
    }
};

template <int TO, int TP>
__global__ __launch_bounds__(kWaveGramThreads) void gram_wave_kernel(
    const float *__restrict__ obs, const float *__restrict__ pred, int64_t N, int mode, float static_dist, int which,
    double *__restrict__ partials) {
    using L = GramLayout<TO, TP>;
    static_assert(TO == 8 && TP == 12, "the MFMA tiling below is laid out for 16 + 24 features");
    constexpr int DO = L::DO, DP = L::DP;
    constexpr int QO = DO / 4, QP = DP / 4;
    constexpr int kWaves = kWaveGramThreads / 64;
    constexpr int PR = QO + QP + 1;  // ONE row per trajectory: [obs 16 | pred 24] + 4 floats of padding = 11 float4 (odd: conflict free)
    static_assert(PR % 2 == 1, "the row pitch must be odd in float4 units");
    constexpr int kSliceF4 = 64 * PR;
    constexpr int kRedDoubles = kWaves * L::kSums;
    constexpr int kLdsDoubles = (2 * kWaves * kSliceF4 > kRedDoubles ? 2 * kWaves * kSliceF4 : kRedDoubles) + 2;
    __shared__ __attribute__((aligned(16))) double sMem[kLdsDoubles];
    double *sRed = sMem;  // reused after the loop
    int *sCountPtr = reinterpret_cast<int *>(sMem + kLdsDoubles - 2);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int kslot = lane >> 4;
    float4 *sRow = reinterpret_cast<float4 *>(sMem) + wave * kSliceF4;
    double acc[L::kPairs];
#pragma unroll
    for (int t = 0; t < L::kPairs; ++t) acc[t] = 0.0;
    // operand lane (i, slot, k) = lane (& 3, >> 2 & 3, >> 4): feature 4 block + i of the trajectory in k-slot k.  The four
    // k-slots of a group take rows {0,3,6,1} / {4,7,2,5} of an 8-row window (dword offsets 44 row + 4 block + i: with the
    // 11-float4 pitch half of a half-wave's ds_read_b32 are 2-way bank conflicted whatever the rows -- an exhaustive search
    // over row assignments and block orders found no conflict-free one, HISTORY.md --, the other half conflict free)
    const int opi = lane & 3, opslot = (lane >> 2) & 3;
    const int rowA = kslot == 0 ? 0 : (kslot == 1 ? 3 : (kslot == 2 ? 6 : 1)), rowB = rowA ^ 4;
    int offA[4], offB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        offA[q] = rowA * (4 * PR) + 4 * gram4_block(opslot, q) + opi;
        offB[q] = rowB * (4 * PR) + 4 * gram4_block(opslot, q) + opi;
    }
    int my_count = 0;
    const int64_t n_tiles = ceil_div(N, (int64_t)64);
    const int64_t first = (int64_t)blockIdx.x * kWaves + wave, stride = (int64_t)gridDim.x * kWaves;
    float4 ro[QO], rp[QP];
    auto fetch = [&](int64_t tile) {
        const int64_t n0 = tile * 64;
        const int rows = (int)min((int64_t)64, N - n0);
        const float4 *go = reinterpret_cast<const float4 *>(obs + n0 * DO);
        const float4 *gp = reinterpret_cast<const float4 *>(pred + n0 * DP);
        if (rows == 64) {
#pragma unroll
            for (int j = 0; j < QO; ++j) ro[j] = ld_rows(go + lane + j * 64);
#pragma unroll
            for (int j = 0; j < QP; ++j) rp[j] = ld_rows(gp + lane + j * 64);
        } else {
#pragma unroll
            for (int j = 0; j < QO; ++j) ro[j] = lane + j * 64 < rows * QO ? ld_rows(go + lane + j * 64) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < QP; ++j) rp[j] = lane + j * 64 < rows * QP ? ld_rows(gp + lane + j * 64) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto wave_sync = [&]() {  // LDS hand-over between the lanes of this wavefront
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    if (first < n_tiles) fetch(first);
    for (int64_t tile = first; tile < n_tiles; tile += stride) {
        const int rows = (int)min((int64_t)64, N - tile * 64);
        wave_sync();  // the previous pass is done with the slice
#pragma unroll
        for (int j = 0; j < QO; ++j) {
            const int q = lane + j * 64;
            sRow[(q / QO) * PR + (q % QO)] = ro[j];
        }
#pragma unroll
        for (int j = 0; j < QP; ++j) {
            const int q = lane + j * 64;
            sRow[(q / QP) * PR + QO + (q % QP)] = rp[j];
        }
        if (tile + stride < n_tiles) fetch(tile + stride);  // in flight during the rest of this pass
        wave_sync();
#if ET_EXP_GRAM >= 2  // measurement aid: loads + stage-in only
        if (0)
#endif
        {
            float4 *orow = sRow + lane * PR, *prow = orow + QO;
            bool use = false;
            if (lane < rows) {
                float xo[DO];
#pragma unroll
                for (int j = 0; j < QO; ++j) {
                    const float4 v = orow[j];
                    xo[4 * j] = v.x;
                    xo[4 * j + 1] = v.y;
                    xo[4 * j + 2] = v.z;
                    xo[4 * j + 3] = v.w;
                }
                const float ox = xo[DO - 2], oy = xo[DO - 1];
                const RowNorm p = row_norm(ox, oy, ox - xo[DO - 6], oy - xo[DO - 5], mode, static_dist);
                use = p.mv == which;
                if (use) {
                    const NormPk np(p);
#pragma unroll
                    for (int j = 0; j < QO; ++j) orow[j] = np.quad(xo[4 * j], xo[4 * j + 1], xo[4 * j + 2], xo[4 * j + 3]);
#pragma unroll
                    for (int q = 0; q < QP; ++q) {
                        const float4 v = prow[q];
                        prow[q] = np.quad(v.x, v.y, v.z, v.w);
                    }
                    ++my_count;
                }
            }
            if (!use) {
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < QO; ++j) orow[j] = z;
#pragma unroll
                for (int q = 0; q < QP; ++q) prow[q] = z;
            }
        }
        wave_sync();
#if ET_EXP_GRAM >= 1  // measurement aid: no matrix phase
        if (0)
#endif
        {
            const float *fb = reinterpret_cast<const float *>(sRow);
#pragma unroll
            for (int g = 0; g < 16; ++g) {  // group g: window g / 2, rows {0,3,6,1} or {4,7,2,5} of it
                const int wo = (g >> 1) * (8 * 4 * PR);
                double v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (double)fb[((g & 1) ? offB[q] : offA[q]) + wo];
#pragma unroll
                for (int qa = 0; qa < 4; ++qa)
#pragma unroll
                    for (int qb = qa; qb < 4; ++qb)
                        acc[gram4_pair(qa, qb)] = __builtin_amdgcn_mfma_f64_4x4x4f64(v[qa], v[qb], acc[gram4_pair(qa, qb)], 0, 0, 0);
            }
        }
    }

    // ---- combine the wavefronts (fixed order), one partial per workgroup
    __syncthreads();  // every wavefront is done with its slice (aliased by sRed)
    if (tid == 0) *sCountPtr = 0;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < L::kPairs; ++t) sRed[wave * L::kSums + t * 64 + lane] = acc[t];
    if (my_count) atomicAdd(sCountPtr, my_count);
    __syncthreads();
    double *dst = partials + (size_t)blockIdx.x * L::kPartial;
    for (int i = tid; i < L::kSums; i += kWaveGramThreads) {
        double sum = 0.0;
        for (int w = 0; w < kWaves; ++w) sum += sRed[w * L::kSums + i];
        dst[i] = sum;
    }
    if (tid == 0) dst[L::kSums] = (double)*sCountPtr;
}

// G (i, j) of the 16 x 16 sub-Gram of slot `s` from the summed accumulators S[pair][lane] (GramLayout): the entry of
// features (4 qa + ia, 4 qb + jb) of the slot's blocks, qa <= qb, sits in pair (qa, qb), lane jb + 4 s + 16 ia.  Every
// entry of the upper triangle is taken from ONE accumulator and mirrored: the matrices are exactly symmetric.
__device__ __forceinline__ double gram4_slot_entry(const double *S, int s, int pa, int ia, int pb, int jb) {
    if (pa > pb || (pa == pb && ia > jb)) {  // the transposed entry
        int t = pa;
        pa = pb;
        pb = t;
        t = ia;
        ia = jb;
        jb = t;
    }
    return S[gram4_pair(pa, pb) * 64 + jb + 4 * s + 16 * ia];
}
__device__ __forceinline__ double gram_obs_entry(const double *S, int i, int j) { return gram4_slot_entry(S, 0, i >> 2, i & 3, j >> 2, j & 3); }
__device__ __forceinline__ double gram_pred_entry(const double *S, int i, int j) {
    if (i > j) {  // (one canonical accumulator per unordered pair: blocks {0,1}, {2,3}, {4,5} are computed by two slots)
        const int t = i;
        i = j;
        j = t;
    }
    const int ba = i >> 2, bb = j >> 2;
    for (int s = 1; s < 4; ++s) {
        int pa = -1, pb = -1;
        for (int q = 0; q < 4; ++q) {
            if (gram4_block(s, q) == 4 + ba) pa = q;
            if (gram4_block(s, q) == 4 + bb) pb = q;
        }
        if (pa >= 0 && pb >= 0) return gram4_slot_entry(S, s, pa, i & 3, pb, j & 3);
    }
    return 0.0;  // (unreachable: the three slots cover every pair of the six blocks)
}

// Sum the workgroup partials of one entry: one wavefront per entry, a fixed strided + butterfly
// order (reproducible for a given grid).  769 small workgroups on purpose: the partials are 4.7 MB, and ONE workgroup
// pulling them through its CU (the reduction folded into the eigensolver's launch, tried in round 6) takes 80 us.
__global__ __launch_bounds__(64) void gram_reduce_kernel(const double *__restrict__ partials, int n_partials, int stride,
                                                         double *__restrict__ sums) {
    const int e = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < n_partials; b += 64) s += partials[(size_t)b * stride + e];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) sums[e] = s;
}

// Assemble the two full symmetric matrices from the three summed 16x16 tiles.  Every entry of the
// upper triangle is taken from exactly one tile and mirrored, so the output is exactly symmetric.
template <int TO, int TP>
__global__ __launch_bounds__(kFitThreads) void gram_finish_kernel(const double *__restrict__ sSum,
                                                                  double *__restrict__ G_obs,
                                                                  double *__restrict__ G_pred,
                                                                  int64_t *__restrict__ count) {
    using L = GramLayout<TO, TP>;
    constexpr int DO = L::DO, DP = L::DP;
    for (int e = threadIdx.x; e < DO * DO; e += kFitThreads) G_obs[e] = gram_obs_entry(sSum, e / DO, e % DO);
    for (int e = threadIdx.x; e < DP * DP; e += kFitThreads) G_pred[e] = gram_pred_entry(sSum, e / DP, e % DP);
    if (threadIdx.x == 0) *count = (int64_t)sSum[L::kSums];
}

// Any-shape Gram: one workgroup per chunk of trajectories, thread = matrix entries.
// fp64 features are staged in LDS per 32-trajectory slab; partials are full matrices.
__global__ __launch_bounds__(kFitThreads) void gram_generic_kernel(
    const float *__restrict__ obs, const float *__restrict__ pred, int64_t N, int T_obs, int T_pred, int mode,
    float static_dist, int which, double *__restrict__ partials) {
    constexpr int kSlab = 32;
    const int DO = 2 * T_obs, DP = 2 * T_pred, D = DO + DP;
    extern __shared__ __attribute__((aligned(16))) double sF[];  // kSlab * D doubles, then one int
    int &sCnt = *reinterpret_cast<int *>(sF + kSlab * D);
    const int tid = threadIdx.x;
    const int n_entries = DO * DO + DP * DP;
    double *dst = partials + (size_t)blockIdx.x * (n_entries + 1);
    for (int e = tid; e < n_entries + 1; e += kFitThreads) dst[e] = 0.0;
    if (tid == 0) sCnt = 0;
    const int64_t n_slabs = ceil_div(N, (int64_t)kSlab);
    for (int64_t slab = blockIdx.x; slab < n_slabs; slab += gridDim.x) {
        const int64_t n0 = slab * kSlab;
        const int rows = (int)min((int64_t)kSlab, N - n0);
        __syncthreads();
        if (tid < kSlab) {
            double *f = sF + tid * D;
            bool use = false;
            if (tid < rows) {
                const float *row = obs + (n0 + tid) * DO;
                const float ox = row[DO - 2], oy = row[DO - 1];
                const RowNorm p = row_norm(ox, oy, ox - row[DO - 6], oy - row[DO - 5], mode, static_dist);
                use = p.mv == which;
                if (use) {
                    for (int t = 0; t < T_obs; ++t) {
                        float a, b;
                        normalize_point(p, row[2 * t], row[2 * t + 1], a, b);
                        f[2 * t] = (double)a;
                        f[2 * t + 1] = (double)b;
                    }
                    const float *prow = pred + (n0 + tid) * DP;
                    for (int t = 0; t < T_pred; ++t) {
                        float a, b;
                        normalize_point(p, prow[2 * t], prow[2 * t + 1], a, b);
                        f[DO + 2 * t] = (double)a;
                        f[DO + 2 * t + 1] = (double)b;
                    }
                    atomicAdd(&sCnt, 1);
                }
            }
            if (!use)
                for (int e = 0; e < D; ++e) f[e] = 0.0;
        }
        __syncthreads();
        for (int e = tid; e < n_entries; e += kFitThreads) {
            int i, j;
            if (e < DO * DO) {
                i = e / DO;
                j = e % DO;
            } else {
                const int r = e - DO * DO;
                i = DO + r / DP;
                j = DO + r % DP;
            }
            double s = dst[e];
            for (int r = 0; r < kSlab; ++r) s = fma(sF[r * D + i], sF[r * D + j], s);
            dst[e] = s;
        }
    }
    __syncthreads();
    if (tid == 0) dst[n_entries] = (double)sCnt;
}

__global__ __launch_bounds__(kFitThreads) void gram_generic_finish_kernel(const double *__restrict__ sums, int DO, int DP,
                                                                          double *__restrict__ G_obs,
                                                                          double *__restrict__ G_pred,
                                                                          int64_t *__restrict__ count) {
    const int n_entries = DO * DO + DP * DP;
    for (int e = threadIdx.x; e < n_entries + 1; e += kFitThreads) {
        const double s = sums[e];
        if (e < DO * DO) G_obs[e] = s;
        else if (e < n_entries) G_pred[e - DO * DO] = s;
        else *count = (int64_t)s;
    }
}

// ------------------------------------------------------------------------------------------
// Parallel-order (round-robin) Jacobi eigensolver in ONE workgroup, fp64, n <= 64, matrix in LDS.
// A round applies n/2 disjoint rotations: lanes compute the (c, s) of one pair each, then every
// 2 x 2 block (pair, pair) of A gets its row rotation followed by its column rotation from ONE lane
// (and V its column rotation): two barriers per round instead of four per rotation.  Same pairing, same formulas and the same per-element
// operation order as the CPU oracle (oracle/et_oracle.c: eto_jacobi) => bit-identical output.
// 24 x 24 converges in ~8 sweeps (23 rounds each).
// (Round 3 tried ONE barrier per round: every work item derives its rotations itself from a double-buffered copy of the
// matrix, no parameter phase -- the same bits, but 239 against 221 us: the fp64 sqrt -> sqrt -> divide chains then run in
// every wavefront instead of one, and that costs more than the barrier and the LDS hand-off it removes.)
// ------------------------------------------------------------------------------------------
constexpr int kJacobiMaxSweeps = 30;
// Stop test at the head of a sweep (oracle/et_oracle.c: eto_jacobi has the same constant).  Jacobi converges
// quadratically: the sweep that finds 1e-10 would leave ~1e-20.  Until round 5 the bound was 1e-15, one more sweep (of
// nine on the bench's matrices); on the 22 Gram matrices of the five splits and the synthetic set the two bounds give
// the same U to 4e-14 (fp32 results: the last bit of a near-zero entry in 3 of 22).
#ifndef ET_EIGH_HEADSTART
#define ET_EIGH_HEADSTART 0
#endif
#ifndef ET_JACOBI_STOP
#define ET_JACOBI_STOP 1e-10
#endif
constexpr double kJacobiStop = ET_JACOBI_STOP;
#ifdef ET_EXP_EIGHSTAMP  // development aid: s_memtime ticks of workgroup 0's first wavefront by phase of a round:
// [0] rounds, [1] block / V items, [2] look-ahead entries, [3] rotation chain + stepping, [4] barrier, [5] sweep checks, [6] sweeps
// (-DET_EXP_EIGHSTAMP=w + 1: wavefront w; 16 = the look-ahead wavefront)
__device__ unsigned long long g_eighstamp[8];
#define ET_EIGHSTAMP(i)                                              \
    do {                                                             \
        if (stamping) {                                              \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
            es_acc[i] += t_ - es_t;                                  \
            es_t = t_;                                               \
        }                                                            \
    } while (0)
#else
#define ET_EIGHSTAMP(i)
#endif
#ifndef ET_EIGH_THREADS
#define ET_EIGH_THREADS 1024
#endif
constexpr int kEighThreads = ET_EIGH_THREADS;  // 16 wavefronts share the element updates of a round: 290 us (256 threads) -> 245 us for 24 x 24; 64 threads: 630 us

// sqrt(x) and 1 / sqrt(x) of a normal positive double to full precision (not correctly rounded): hardware estimate
// (v_rsq_f64) + two coupled Goldschmidt steps, ~10 dependent fp64 operations instead of two ~25-operation IEEE sequences.
// No range scaling: the arguments here are sums of squares of Gram-matrix entries (1e-40 ... 1e+30).
__device__ __forceinline__ void sqrt_rsqrt(double x, double &root, double &rroot) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    root = g;
    rroot = h + h;
}

// pair i of round r of the round-robin schedule on m (even) players -> (p, q), p < q  (oracle/et_oracle.c: eto_jacobi)
__device__ __forceinline__ void jacobi_schedule(int m, int r, int i, int &p, int &q) {
    int a, b;
    if (i == 0) {
        a = m - 1;
        b = r;
    } else {
        a = r + i;  // (r + i) % (m - 1) with r, i < m - 1
        a = a >= m - 1 ? a - (m - 1) : a;
        b = r + (m - 1) - i;
        b = b >= m - 1 ? b - (m - 1) : b;
    }
    p = a < b ? a : b;
    q = a < b ? b : a;
}

// The same schedule carried from round to round: every position but the fixed player's moves on by one modulo m - 1 each
// round (and is back where it started after the m - 1 rounds of a sweep), so a work item keeps the positions of its pairs
// in registers and advances them with add / subtract / unsigned-min -- no compare -> select chains through VCC.  (Deriving
// the pairs from the round number cost 340 of the 980 cycles of an update, profiles/r04i_eigh_stamps.txt.)
struct JacobiPair {
    int a, b;
    bool fixed;  // pair 0: its first player is m - 1 in every round
    __device__ __forceinline__ void init(int m, int i) {
        fixed = i == 0;
        a = i;
        b = i == 0 ? 0 : m - 1 - i;
    }
    __device__ __forceinline__ void get(int m, int &p, int &q) const {
        const int ae = fixed ? m - 1 : a;
        p = ae < b ? ae : b;
        q = ae < b ? b : ae;
    }
    __device__ __forceinline__ void step(int m) {
        const unsigned a1 = (unsigned)a + 1u, b1 = (unsigned)b + 1u, w = (unsigned)(m - 1);
        a = (int)min(a1, a1 - w);
        b = (int)min(b1, b1 - w);
    }
};

// `gload(r, c)`: entry (r, c) of the matrix (from memory, or from LDS tiles in the fused fit kernel); `sm`: the body's LDS
// (eigh_lds_bytes(n) bytes, 16-byte aligned).
template <class GLoad>
__device__ __forceinline__ void eigh_topk_body(GLoad gload, int n, int k, float *__restrict__ U, float *__restrict__ sigma,
                                               double *sm) {
    // Round 4: the matrix lives in LDS padded with zeros to m x m (m = n rounded up to even).  The padding index pairs
    // with a zero entry, i.e. is an inactive pair, and an INACTIVE pair is applied as the identity rotation (c, s) = (1, 0)
    // -- 1 * x - 0 * y is x again, bit for bit up to the sign of a zero -- so the update phase has no `q < n` / `active`
    // branches: every work item reads its four entries in one go, rotates twice, writes four; the maxima of the stop test
    // go through two LDS integer maxima (|x| of non-negative doubles orders like its bit pattern).
    //
    // Round 6: ONE barrier per round instead of two.  A round used to be  parameters (12 lanes: LDS read -> the dependent
    // fp64 chain alpha, beta -> rsq + Goldschmidt twice -> c, s; 644 cycles) | barrier | updates (760 cycles) | barrier
    // (profiles/r04i_eigh_stamps.txt) -- the sum of the two phases, ~200 times per solve.  Now the parameter lanes sit in
    // a wavefront of their own (the last one) and run one round AHEAD, beside the updates: the three entries a_pq, a_pp,
    // a_qq that the NEXT round's pair (p, q) will need are each one entry of a 2 x 2 block this round's updates rewrite
    // -- block (pair of p, pair of q), and the two diagonal blocks --, so the parameter lane evaluates exactly those
    // three entries itself, with the update items' own formulas on the same inputs (row rotation of the pair holding the
    // row, then column rotation of the pair holding the column: the same products and sums in the same order, hence the
    // same bits), and goes straight on to the next round's c, s.  The matrix is double-buffered (every entry is rewritten
    // by exactly one block item per round: read A[cur], write A[cur ^ 1]) so that look-ahead reads and update writes do
    // not race; so are the parameters.  A round is max(updates, look-ahead + chain) + one barrier.  Same arithmetic per
    // element as before -- tools/eigh_accuracy.py and the oracle comparison are unchanged.
    const int lane = threadIdx.x;
    const int m = (n + 1) & ~1, half = m / 2;
    double *Abuf = sm;               // 2 * m*m
    double *V = sm + 2 * m * m;      // m*m
    double *sC = V + m * m;          // [2][32] cosines
    double *sS = sC + 64;            // [2][32] sines
    unsigned long long *sMax = reinterpret_cast<unsigned long long *>(sS + 64);  // [sweep parity][off, diag]
    int *sAct = reinterpret_cast<int *>(sMax + 4);  // [2][32] active, 64 used
    int *sUsed = sAct + 64;
    uint2 *sTab = reinterpret_cast<uint2 *>(sUsed + 64);  // [m - 1 rounds][3 half look-ahead items]
    for (int i = lane; i < m * m; i += kEighThreads) {
        const int r = i / m, c = i - r * m;
        Abuf[i] = (r < n && c < n) ? gload(r, c) : 0.0;
        V[i] = (r == c && r < n) ? 1.0 : 0.0;
    }
    if (lane < 4) sMax[lane] = 0ull;
    // Look-ahead table (see "Round 6" above): item (e, L) of round r describes ONE entry of the matrix after round r that
    // parameter lane L needs for its pair (p, q) of round r + 1 -- e = 0: a_pq, 1: a_pp, 2: a_qq -- as the 2 x 2 block of
    // round r it lies in: the four element offsets, the pairs whose rotations act on its rows (iR) and columns (iC), and
    // whether the entry is the block's second row / second column.  Round-robin schedules repeat every m - 1 rounds.
    const int nItems = 3 * half;
    for (int idx = lane; idx < (m - 1) * nItems; idx += kEighThreads) {
        const int r = idx / nItems, it = idx - r * nItems, e = it / half, Lt = it - e * half;
        int p, q;
        jacobi_schedule(m, r + 1 == m - 1 ? 0 : r + 1, Lt, p, q);
        int i1 = 0, p1 = 0, q1 = 0, i2 = 0, p2 = 0, q2 = 0;
        for (int i = 0; i < half; ++i) {
            int pi, qi;
            jacobi_schedule(m, r, i, pi, qi);
            if (p == pi || p == qi) { i1 = i; p1 = pi; q1 = qi; }
            if (q == pi || q == qi) { i2 = i; p2 = pi; q2 = qi; }
        }
        const int iR = e == 2 ? i2 : i1, pR = e == 2 ? p2 : p1, qR = e == 2 ? q2 : q1;
        const int iC = e == 1 ? i1 : i2, pC = e == 1 ? p1 : p2, qC = e == 1 ? q1 : q2;
        const unsigned row2 = (e == 2 ? q : p) == qR, col2 = (e == 1 ? p : q) == qC, zero = e == 0 && i1 == i2;
        sTab[idx] = make_uint2((unsigned)(pR * m + pC) | (unsigned)(pR * m + qC) << 12 | (unsigned)iR << 24,
                               (unsigned)(qR * m + pC) | (unsigned)(qR * m + qC) << 12 | (unsigned)iC << 24 | row2 << 29 |
                                   col2 << 30 | zero << 31);
    }
    __syncthreads();
    // work items of this thread in the update phase:
    //   V: e = lane + t*T - T/2 -> (pair i, row j): columns p_i, q_i of row j
    //   A: b = lane + t*T       -> (pair i1, pair i2): the 2 x 2 block rows {p1, q1} x columns {p2, q2}
    // (the V items start half a workgroup away from the A blocks: for the usual small n the two kinds of work land on
    // different wavefronts and a round's critical path is the longer of the two, not their sum)
    // (round 6) ... and on wavefronts that do not share a SIMD with the look-ahead wavefront: a workgroup's wavefronts go to
    // the four SIMDs cyclically, w and w + 4 sit on the same one, and two busy wavefronts on a SIMD take turns issuing --
    // the V items on wavefront 11 doubled the latency of the chain on wavefront 15.  V items: wavefronts 4 5 6, 8 9 10,
    // 12 13 14 (in this order), 576 per slot.
    constexpr bool kSpread = kEighThreads == 1024;
    constexpr int kVShift = kEighThreads / 2;
    constexpr int kVPerSlot = kSpread ? 9 * 64 : kEighThreads;
    constexpr int kSlots = kSpread ? (32 * 64 + kVPerSlot - 1) / kVPerSlot : (32 * 64 + kVShift + kEighThreads - 1) / kEighThreads;
    constexpr int kBlkSlots = (32 * 32 + kEighThreads - 1) / kEighThreads;
    constexpr int kChkSlots = (64 * 64 + kEighThreads - 1) / kEighThreads;
    constexpr int kParBase = kEighThreads >= 128 ? kEighThreads - 64 : 0;  // the parameter lanes: first lanes of the last wavefront
    int slot_i[kSlots], slot_row[kSlots], blk_1[kBlkSlots], blk_2[kBlkSlots], chk[kChkSlots];
#pragma unroll
    for (int t = 0; t < kSlots; ++t) {
        int e;
        if (kSpread) {
            const int w = lane >> 6;
            e = (w >= 4 && (w & 3) != 3) ? ((w - 4) - ((w - 4) >> 2)) * 64 + (lane & 63) + t * kVPerSlot : -1;
        } else {
            e = lane + t * kEighThreads - kVShift;
        }
        const bool ok = e >= 0 && e < half * n;
        slot_i[t] = ok ? e / n : -1;
        slot_row[t] = ok ? (e % n) * m : 0;  // (row j) * m
    }
#pragma unroll
    for (int t = 0; t < kBlkSlots; ++t) {
        const int b = lane + t * kEighThreads;
        blk_1[t] = b < half * half ? b / half : -1;
        blk_2[t] = b < half * half ? b % half : 0;
    }
    // (carried positions for the parameter lanes and the A blocks -- the critical path; the V items, on other wavefronts
    // and shorter, derive theirs from the round number: stepping three more slots on every wavefront costs what it saves)
    JacobiPair b1[kBlkSlots], b2[kBlkSlots];
#pragma unroll
    for (int t = 0; t < kBlkSlots; ++t) {
        b1[t].init(m, blk_1[t] < 0 ? 0 : blk_1[t]);
        b2[t].init(m, blk_2[t]);
    }
    // The look-ahead items live in the LAST wavefront.  4 half <= 64 (n <= 32): one item per lane, a quad of lanes per
    // pair -- lane 4 L holds a_pq and goes on to the rotation, lanes 4 L + 1 / + 2 hold a_pp / a_qq and hand them over with
    // DPP quad broadcasts; larger n: lane L < half evaluates its three items one after the other.
    const int pl = lane - kParBase;
    const bool split = 4 * half <= 64;
    const bool is_par = pl >= 0 && (split ? (pl & 3) == 0 && (pl >> 2) < half : pl < half);  // runs the rotation chain, writes c / s
    const int L = is_par ? (split ? pl >> 2 : pl) : 0;
    // split: lane 4 L + e holds item (e, L), e = 0, 1, 2 (the quad's fourth lane idles); serial: lane L holds items (0..2, L)
    const int item0 = pl < 0 ? 0 : (split ? ((pl & 3) < 3 && (pl >> 2) < half ? (pl & 3) * half + (pl >> 2) : 0) : (pl < half ? pl : 0));
    JacobiPair par;
    par.init(m, L);
#pragma unroll
    for (int t = 0; t < kChkSlots; ++t) {  // element of the upper triangle or the diagonal: its LDS index, +(1 << 20) for the diagonal
        const int e = lane + t * kEighThreads;
        const int i = e / n, j = e - i * n;
        chk[t] = (e < n * n && j >= i) ? (i * m + j) | (i == j ? 1 << 20 : 0) : -1;
    }
    // c = D / g, s = sgn |beta| / g with D = |alpha| + sqrt(alpha^2 + beta^2), g = sqrt(D^2 + beta^2) (the oracle's
    // formulas).  This chain -- sqrt -> sqrt -> divide, each a ~100-150 ns correctly rounded software sequence in fp64 --
    // is on the critical path of every one of the ~200 rounds; here both roots come from v_rsq_f64 + two Goldschmidt
    // steps (sqrt_rsqrt: full double precision, not correctly rounded) and the divisions become multiplications by
    // 1 / g.  c^2 + s^2 = 1 to a few 1e-16; the result is not bit-identical to the oracle's correctly rounded chain (U
    // agrees to ~1e-14, i.e. to the last bit of its fp32 value except on a rounding boundary) but is the same on every
    // GPU / rank.  An inactive pair (a_pq == 0: also the padding pair of an odd n) runs the chain on zeros and drops it.
    auto rotation = [&](double apq, double app, double aqq, int buf) {
        const double alpha = aqq - app, beta = 2.0 * apq;
#ifdef ET_EIGH_IEEE_PARAMS
        const double h = sqrt(alpha * alpha + beta * beta);
        const double D = fabs(alpha) + h;
        const double g = sqrt(D * D + beta * beta);
        const double rg = 1.0 / g;
#else
        double h, rh;
        sqrt_rsqrt(alpha * alpha + beta * beta, h, rh);
        const double D = fabs(alpha) + h;
        double g, rg;
        sqrt_rsqrt(D * D + beta * beta, g, rg);
#endif
        const double sgn = (alpha == 0.0 || ((alpha > 0.0) == (beta > 0.0))) ? 1.0 : -1.0;
        const bool act = apq != 0.0;
        sC[buf * 32 + L] = act ? D * rg : 1.0;
        sS[buf * 32 + L] = act ? sgn * fabs(beta) * rg : 0.0;
        sAct[buf * 32 + L] = act ? 1 : 0;
    };
    int cur = 0, pb = 0;
    if (pl >= 0) __builtin_amdgcn_s_setprio(3);  // the look-ahead wavefront is every round's critical path
    if (is_par) {  // the parameters of the very first round, from the matrix itself
        int p, q;
        par.get(m, p, q);
        rotation(Abuf[__mul24(p, m) + q], Abuf[__mul24(p, m) + p], Abuf[__mul24(q, m) + q], 0);
    }
    uint2 ent[3] = {sTab[item0], sTab[split ? item0 : half + item0], sTab[split ? item0 : 2 * half + item0]};  // round 0's items
#ifdef ET_EXP_EIGHSTAMP
    const bool stamping = n == 24 && (int)(threadIdx.x >> 6) == ET_EXP_EIGHSTAMP - 1;  // wavefront ET_EXP_EIGHSTAMP - 1 (16: the look-ahead one)
    unsigned long long es_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, es_t = __builtin_amdgcn_s_memtime();
#endif
    for (int sweep = 0; sweep < kJacobiMaxSweeps; ++sweep) {
#ifdef ET_EXP_EIGHSTAMP
        if (stamping) { es_t = __builtin_amdgcn_s_memtime(); es_acc[6] += 1; }
#endif
        // converged when max |off-diagonal| <= 1e-10 max |diagonal| (maxima: order independent)
        const double *Ac = Abuf + cur * m * m;
        unsigned long long *mx = sMax + 2 * (sweep & 1);
#pragma unroll
        for (int t = 0; t < kChkSlots; ++t) {
            if (chk[t] < 0) continue;
            const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(fabs(Ac[chk[t] & 0xfffff])));
            atomicMax(mx + (chk[t] >> 20), bits);
        }
        if (lane < 2) sMax[2 * ((sweep + 1) & 1) + lane] = 0ull;  // (the other parity: last read a sweep of barriers ago)
        __syncthreads();  // (also publishes the first round's parameters)
        const double off = __longlong_as_double(static_cast<long long>(mx[0])), diag = __longlong_as_double(static_cast<long long>(mx[1]));
        ET_EIGHSTAMP(5);
        if (off <= kJacobiStop * diag) break;
        for (int r = 0; r < m - 1; ++r) {
#ifdef ET_EXP_EIGHSTAMP
            if (stamping) es_acc[0] += 1;
#endif
#if ET_EIGH_HEADSTART > 0
            // the look-ahead wavefront's LDS reads go first: right after the barrier all sixteen wavefronts send theirs, and
            // the reads the critical path waits for stood in that queue (profiles/r06c_eigh_stamps.txt: 900 of a round's 1 350
            // ticks); the others have ~500 ticks of slack per round
            if (pl < 0) __builtin_amdgcn_s_sleep(ET_EIGH_HEADSTART);
#endif
            const double *Ar = Abuf + cur * m * m;
            double *Aw = Abuf + (cur ^ 1) * m * m;
            const double *pC = sC + pb * 32, *pS = sS + pb * 32;
            const int *pAct = sAct + pb * 32;
            if (pl >= 0) {  // (the whole last wavefront takes the branch: the shuffles below need every lane)
                // look-ahead: entry (row x, column y) of the matrix AFTER this round, x in this round's pair iR = (pR, qR),
                // y in pair iC = (pC, qC):  t_c = alR A[pR][c] + beR A[qR][c] for c = pC, qC with (alR, beR) = (c, -s) of
                // pair iR for the pair's first row, (s, c) for its second (c x - s y and s x + c y of the block items:
                // (-s) y rounds to -(s y), the sum is the same), then alC t_pC + beC t_qC likewise with pair iC.
                auto entry = [&](uint2 t) -> double {
                    const double x_pp = Ar[t.x & 0xfffu], x_pq = Ar[(t.x >> 12) & 0xfffu];
                    const double x_qp = Ar[t.y & 0xfffu], x_qq = Ar[(t.y >> 12) & 0xfffu];
                    const int iR = (int)((t.x >> 24) & 31u), iC = (int)((t.y >> 24) & 31u);
                    const double cR = pC[iR], sR = pS[iR], cC = pC[iC], sC_ = pS[iC];
                    const int actR = pAct[iR];
                    const bool row2 = (t.y >> 29) & 1u, col2 = (t.y >> 30) & 1u;
                    const double alR = row2 ? sR : cR, beR = row2 ? cR : -sR;
                    const double alC = col2 ? sC_ : cC, beC = col2 ? cC : -sC_;
                    const double v = alC * (alR * x_pp + beR * x_qp) + beC * (alR * x_pq + beR * x_qq);
                    // (m = 2: the next pair IS this pair; its rotated off-diagonal entry is exactly zero, like the block item's)
                    return ((t.y >> 31) & (unsigned)actR) ? 0.0 : v;
                };
                double apq, app, aqq;
                if (split) {  // a_pp, a_qq from the quad's lanes 1 and 2: DPP moves, no trip through the LDS
                    const double v = entry(ent[0]);
                    const int lo = __double2loint(v), hi = __double2hiint(v);
                    apq = v;
                    app = __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x55, 0xf, 0xf, false), __builtin_amdgcn_mov_dpp(lo, 0x55, 0xf, 0xf, false));
                    aqq = __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0xaa, 0xf, 0xf, false), __builtin_amdgcn_mov_dpp(lo, 0xaa, 0xf, 0xf, false));
                } else {
                    apq = entry(ent[0]);
                    app = entry(ent[1]);
                    aqq = entry(ent[2]);
                }
                ET_EIGHSTAMP(2);  // [2] look-ahead entries (+ shuffles)
                // the next round's items: in flight during the chain below
                const int rn = r + 1 == m - 1 ? 0 : r + 1;
                ent[0] = sTab[rn * nItems + item0];
                if (!split) {
                    ent[1] = sTab[rn * nItems + half + item0];
                    ent[2] = sTab[rn * nItems + 2 * half + item0];
                }
                if (is_par) rotation(apq, app, aqq, pb ^ 1);
            }
            ET_EIGHSTAMP(3);  // [3] rotation chain
            // A' = J^T A J for the round's disjoint rotations, one 2 x 2 block per work item: the row rotation of pair
            // i1 followed by the column rotation of pair i2 touches exactly these four entries, so "all row updates,
            // then all column updates" (the oracle's order, with its intermediate roundings) needs no barrier in between.
#pragma unroll
            for (int t = 0; t < kBlkSlots; ++t) {
                const int i1 = blk_1[t], i2 = blk_2[t];
                if (i1 < 0) continue;
                // (the pairs follow from the round number, so the rotations AND the four matrix entries are requested from
                // LDS together: one round trip after the barrier, not two)
                int p1, q1, p2, q2;
                b1[t].get(m, p1, q1);
                b2[t].get(m, p2, q2);
                const int rp = __mul24(p1, m), rq = __mul24(q1, m);
                const int diag_act = pAct[i1] & (i1 == i2 ? 1 : 0);  // (read by every lane: a read under a branch is a round trip of its own)
                const double c1 = pC[i1], s1 = pS[i1], c2 = pC[i2], s2 = pS[i2];
                const double x_pp = Ar[rp + p2], x_pq = Ar[rp + q2], x_qp = Ar[rq + p2], x_qq = Ar[rq + q2];
                // rows p1, q1 (columns p2 and q2), then columns p2, q2 (rows p1 and q1)
                const double t_pp = c1 * x_pp - s1 * x_qp, t_qp = s1 * x_pp + c1 * x_qp;
                const double t_pq = c1 * x_pq - s1 * x_qq, t_qq = s1 * x_pq + c1 * x_qq;
                const double r_pp = c2 * t_pp - s2 * t_pq, r_qp = c2 * t_qp - s2 * t_qq;
                double r_pq = s2 * t_pp + c2 * t_pq, r_qp2 = r_qp;
                const double r_qq = s2 * t_qp + c2 * t_qq;
                if (diag_act) {  // the rotated pair entries are exactly zero, like in the oracle
                    r_pq = 0.0;
                    r_qp2 = 0.0;
                }
                Aw[rp + p2] = r_pp;
                Aw[rp + q2] = r_pq;
                Aw[rq + p2] = r_qp2;
                Aw[rq + q2] = r_qq;
            }
#pragma unroll
            for (int t = 0; t < kSlots; ++t) {  // V' = V J: columns p, q of every row
                const int i = slot_i[t];
                if (i < 0) continue;
                int p, q;
                jacobi_schedule(m, r, i, p, q);
                const double c = pC[i], sn = pS[i];
                double *row = V + slot_row[t];
                const double vjp = row[p], vjq = row[q];
                row[p] = c * vjp - sn * vjq;
                row[q] = sn * vjp + c * vjq;
            }
#pragma unroll
            for (int t = 0; t < kBlkSlots; ++t) {
                b1[t].step(m);
                b2[t].step(m);
            }
            ET_EIGHSTAMP(1);  // [1] this wavefront's block / V items
            __syncthreads();
            ET_EIGHSTAMP(4);
            cur ^= 1;
            pb ^= 1;
        }
    }
    double *A = Abuf + cur * m * m;
#ifdef ET_EXP_EIGHSTAMP
    if (stamping && (threadIdx.x & 63) == 0)
        for (int i = 0; i < 7; ++i) atomicAdd(&g_eighstamp[i], es_acc[i]);
#endif
    __syncthreads();
    // Top-k extraction, all columns at once (a single lane walking through n diagonal entries and n vector entries per
    // column cost ~4 us per column: 22 us of a 243 us solve).  Rank of eigenvalue i = number of eigenvalues that come before
    // it in "largest first, lower index first on ties" order -- the order the serial selection (strict `>`) produces; the
    // sign makes the first entry of largest magnitude positive.
    if (lane < n) {
        const double di = A[lane * m + lane];
        int rank = 0;
        for (int i = 0; i < n; ++i) {
            const double dj = A[i * m + i];
            rank += (dj > di || (dj == di && i < lane)) ? 1 : 0;
        }
        sUsed[lane] = rank;
    }
    __syncthreads();
    for (int e = lane; e < n * k; e += kEighThreads) {
        const int col = e / k, j = e - col * k;  // work item: eigenvector `col`, if its rank is j
        if (sUsed[col] != j) continue;
        int im = 0;
        for (int i = 1; i < n; ++i)
            if (fabs(V[i * m + col]) > fabs(V[im * m + col])) im = i;
        const double sgn = V[im * m + col] < 0.0 ? -1.0 : 1.0;
        const double lam = A[col * m + col];
        sigma[j] = (float)sqrt(lam > 0.0 ? lam : 0.0);
        for (int i = 0; i < n; ++i) U[i * k + j] = (float)(sgn * V[i * m + col]);
    }
}

__global__ __launch_bounds__(kEighThreads) void eigh_topk_kernel(const double *__restrict__ G, int n, int k,
                                                                  float *__restrict__ U, float *__restrict__ sigma) {
    extern __shared__ __attribute__((aligned(16))) double eigh_smem[];
    eigh_topk_body([=](int r, int c) { return G[r * n + c]; }, n, k, U, sigma, eigh_smem);
}

// several independent matrices in one launch, one workgroup each (the obs / pred, moving / static Gram
// matrices of a fit are solved side by side instead of one after the other)
struct EighBatch {
    const double *G[ET_EIGH_MAX_BATCH];
    float *U[ET_EIGH_MAX_BATCH];
    float *sigma[ET_EIGH_MAX_BATCH];
    int n[ET_EIGH_MAX_BATCH];
    int k[ET_EIGH_MAX_BATCH];
};

__global__ __launch_bounds__(kEighThreads) void eigh_topk_batch_kernel(const EighBatch b) {
    extern __shared__ __attribute__((aligned(16))) double eigh_smem[];
    const int i = blockIdx.x;
    const double *G = b.G[i];
    const int n = b.n[i];
    eigh_topk_body([=](int r, int c) { return G[r * n + c]; }, n, b.k[i], b.U[i], b.sigma[i], eigh_smem);
}

// The fit of ONE descriptor behind the Gram kernel and its partial reduction: workgroup 0 assembles G_obs (16 x 16) from
// the summed obs tile and solves it, workgroup 1 G_pred (24 x 24) from the two pred tiles -- gram_finish_kernel's assembly
// folded into the eigensolver's launch (one launch and one kernel boundary fewer: -8 us of a 0.5 ms fit).
// G_obs / G_pred / count are optional outputs (the same bits et_fit_gram writes).
template <int TO, int TP>
__global__ __launch_bounds__(kEighThreads) void fit_finish_eigh_kernel(const double *__restrict__ sums, int k,
                                                                       double *__restrict__ G_obs, double *__restrict__ G_pred,
                                                                       int64_t *__restrict__ count, float *__restrict__ U_obs,
                                                                       float *__restrict__ U_pred, float *__restrict__ sigma_obs,
                                                                       float *__restrict__ sigma_pred) {
    using L = GramLayout<TO, TP>;
    constexpr int DO = L::DO, DP = L::DP;
    extern __shared__ __attribute__((aligned(16))) double eigh_smem[];
    double *sT = eigh_smem;  // the summed accumulators (GramLayout::kSums <= 768 doubles)
    const bool is_pred = blockIdx.x == 1;
    for (int e = threadIdx.x; e < L::kSums; e += kEighThreads) sT[e] = sums[e];
    if (!is_pred && count && threadIdx.x == 0) *count = (int64_t)sums[L::kSums];
    __syncthreads();
    if (!is_pred) {
        if (G_obs)
            for (int e = threadIdx.x; e < DO * DO; e += kEighThreads) G_obs[e] = gram_obs_entry(sT, e / DO, e % DO);
        eigh_topk_body([=](int r, int c) { return gram_obs_entry(sT, r, c); }, DO, k, U_obs, sigma_obs, eigh_smem + 768);
    } else {
        if (G_pred)
            for (int e = threadIdx.x; e < DP * DP; e += kEighThreads) G_pred[e] = gram_pred_entry(sT, e / DP, e % DP);
        eigh_topk_body([=](int r, int c) { return gram_pred_entry(sT, r, c); }, DP, k, U_pred, sigma_pred, eigh_smem + 768);
    }
}

static int fit_grid(int64_t N) {
    // one resident round of workgroups (3 per CU at 48 KB of LDS: four wavefront slices), few enough that the
    // partial reduction stays trivial
    const int64_t tiles = ceil_div(N, (int64_t)kWaveGramThreads);
    return (int)(tiles < 768 ? (tiles > 0 ? tiles : 1) : 768);
}

}  // namespace et

using namespace et;

extern "C" size_t et_fit_gram_workspace_bytes(int64_t N, int T_obs, int T_pred) {
    const size_t DO = 2 * (size_t)T_obs, DP = 2 * (size_t)T_pred;
    const size_t per = DO * DO + DP * DP + 1;  // the generic layout is the larger one
    return sizeof(double) * per * ((size_t)fit_grid(N) + 1);  // workgroup partials + their sums
}

extern "C" size_t et_fit_descriptor_workspace_bytes(int64_t N, int T_obs, int T_pred) {
    const size_t DO = 2 * (size_t)T_obs, DP = 2 * (size_t)T_pred;
    return et_fit_gram_workspace_bytes(N, T_obs, T_pred) + sizeof(double) * (DO * DO + DP * DP + 2);  // + G_obs, G_pred, count
}

extern "C" int et_fit_gram(const float *obs, const float *pred, int64_t N, int T_obs, int T_pred, int mode,
                           float static_dist, int which, double *G_obs, double *G_pred, int64_t *count,
                           void *workspace, size_t workspace_bytes, et_stream_t stream) {
    if (N < 0 || T_obs < 3 || T_obs > ET_MAX_T || T_pred < 1 || T_pred > ET_MAX_T || mode < 0 || mode > 3 ||
        (which != 0 && which != 1) || !G_obs || !G_pred || !count)
        return ET_ERR_INVALID_ARG;
    if (N > 0 && (!obs || !pred)) return ET_ERR_INVALID_ARG;
    if (workspace_bytes < et_fit_gram_workspace_bytes(N, T_obs, T_pred) || !workspace) return ET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int DO = 2 * T_obs, DP = 2 * T_pred;
    if (N == 0) {
        ET_HIP_TRY(hipMemsetAsync(G_obs, 0, sizeof(double) * DO * DO, st));
        ET_HIP_TRY(hipMemsetAsync(G_pred, 0, sizeof(double) * DP * DP, st));
        ET_HIP_TRY(hipMemsetAsync(count, 0, sizeof(int64_t), st));
        return ET_OK;
    }
    const int grid = fit_grid(N);
    double *partials = (double *)workspace;
    if (T_obs == 8 && T_pred == 12 && aligned16(obs) && aligned16(pred)) {
        hipLaunchKernelGGL((gram_wave_kernel<8, 12>), dim3(grid), dim3(kWaveGramThreads), 0, st, obs, pred, N, mode,
                           static_dist, which, partials);
        ET_LAUNCH_CHECK();
        constexpr int per = GramLayout<8, 12>::kPartial;
        double *sums = partials + (size_t)grid * per;
        hipLaunchKernelGGL(gram_reduce_kernel, dim3(per), dim3(64), 0, st, partials, grid, per, sums);
        ET_LAUNCH_CHECK();
        hipLaunchKernelGGL((gram_finish_kernel<8, 12>), dim3(1), dim3(kFitThreads), 0, st, sums, G_obs, G_pred, count);
    } else {
        const size_t lds = sizeof(double) * (32 * (size_t)(DO + DP) + 1);
        hipLaunchKernelGGL(gram_generic_kernel, dim3(grid), dim3(kFitThreads), lds, st, obs, pred, N, T_obs, T_pred,
                           mode, static_dist, which, partials);
        ET_LAUNCH_CHECK();
        const int per = DO * DO + DP * DP + 1;
        double *sums = partials + (size_t)grid * per;
        hipLaunchKernelGGL(gram_reduce_kernel, dim3(per), dim3(64), 0, st, partials, grid, per, sums);
        ET_LAUNCH_CHECK();
        hipLaunchKernelGGL(gram_generic_finish_kernel, dim3(1), dim3(kFitThreads), 0, st, sums, DO, DP, G_obs, G_pred,
                           count);
    }
    ET_LAUNCH_CHECK();
    return ET_OK;
}

static size_t eigh_lds_bytes(int n) {
    const size_t m = ((size_t)n + 1) & ~(size_t)1;  // zero-padded to even
    // A (double-buffered), V, c / s (x2), maxima; active (x2), used; the look-ahead table
    return sizeof(double) * (3 * m * m + 128 + 4) + sizeof(int) * (64 + 64) + sizeof(uint2) * (m - 1) * 3 * (m / 2);
}

extern "C" int et_eigh_topk_batch(int batch, const double *const *G, const int *n, const int *k, float *const *U,
                                  float *const *sigma, et_stream_t stream) {
    if (batch < 0 || batch > ET_EIGH_MAX_BATCH || (batch > 0 && (!G || !n || !k || !U || !sigma))) return ET_ERR_INVALID_ARG;
    if (batch == 0) return ET_OK;
    EighBatch b;
    size_t lds = 0;
    for (int i = 0; i < batch; ++i) {
        if (!G[i] || !U[i] || !sigma[i] || n[i] < 1 || n[i] > 64 || k[i] < 1 || k[i] > n[i]) return ET_ERR_INVALID_ARG;
        b.G[i] = G[i];
        b.U[i] = U[i];
        b.sigma[i] = sigma[i];
        b.n[i] = n[i];
        b.k[i] = k[i];
        const size_t need = eigh_lds_bytes(n[i]);
        lds = need > lds ? need : lds;
    }
    if (lds > 48 * 1024)
        ET_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(eigh_topk_batch_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(eigh_topk_batch_kernel, dim3(batch), dim3(kEighThreads), lds, (hipStream_t)stream, b);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_eigh_topk(const double *G, int n, int k, float *U, float *sigma, et_stream_t stream) {
    if (!G || !U || !sigma || n < 1 || n > 64 || k < 1 || k > n) return ET_ERR_INVALID_ARG;
    const size_t lds = eigh_lds_bytes(n);
    if (lds > 48 * 1024)
        ET_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(eigh_topk_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(eigh_topk_kernel, dim3(1), dim3(kEighThreads), lds, (hipStream_t)stream, G, n, k, U, sigma);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_fit_descriptor(const float *obs, const float *pred, int64_t N, int T_obs, int T_pred, int k, int mode,
                                 float static_dist, int which, float *U_obs, float *U_pred, float *sigma_obs,
                                 float *sigma_pred, double *G_obs, double *G_pred, int64_t *count, void *workspace,
                                 size_t workspace_bytes, et_stream_t stream) {
    if (!U_obs || !U_pred || !sigma_obs || !sigma_pred || k < 1 || T_obs < 3 || T_obs > ET_MAX_T || T_pred < 1 || T_pred > ET_MAX_T ||
        k > 2 * T_obs || k > 2 * T_pred)
        return ET_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool fused = N > 0 && T_obs == 8 && T_pred == 12 && obs && pred && aligned16(obs) && aligned16(pred) && mode >= 0 &&
                       mode <= 3 && (which == 0 || which == 1);
    if (!fused) {  // any other shape (and N = 0): the two calls one after the other; G goes through the workspace's tail
        const size_t gram_ws = et_fit_gram_workspace_bytes(N, T_obs, T_pred);
        const size_t DO = 2 * (size_t)T_obs, DP = 2 * (size_t)T_pred;
        if (!workspace || workspace_bytes < et_fit_descriptor_workspace_bytes(N, T_obs, T_pred)) return ET_ERR_WORKSPACE;
        double *tail = reinterpret_cast<double *>(static_cast<char *>(workspace) + gram_ws);
        double *go = G_obs ? G_obs : tail, *gp = G_pred ? G_pred : tail + DO * DO;
        int64_t *cn = count ? count : reinterpret_cast<int64_t *>(tail + DO * DO + DP * DP);
        const int rc = et_fit_gram(obs, pred, N, T_obs, T_pred, mode, static_dist, which, go, gp, cn, workspace, gram_ws, stream);
        if (rc) return rc;
        const double *Gs[2] = {go, gp};
        const int ns[2] = {(int)DO, (int)DP}, ks[2] = {k, k};
        float *Us[2] = {U_obs, U_pred}, *ss[2] = {sigma_obs, sigma_pred};
        return et_eigh_topk_batch(2, Gs, ns, ks, Us, ss, stream);
    }
    if (!workspace || workspace_bytes < et_fit_descriptor_workspace_bytes(N, T_obs, T_pred)) return ET_ERR_WORKSPACE;
    const int grid = fit_grid(N);
    double *partials = (double *)workspace;
    hipLaunchKernelGGL((gram_wave_kernel<8, 12>), dim3(grid), dim3(kWaveGramThreads), 0, st, obs, pred, N, mode, static_dist, which,
                       partials);
    ET_LAUNCH_CHECK();
    constexpr int per = GramLayout<8, 12>::kPartial;
    double *sums = partials + (size_t)grid * per;
    hipLaunchKernelGGL(gram_reduce_kernel, dim3(per), dim3(64), 0, st, partials, grid, per, sums);
    ET_LAUNCH_CHECK();
    const size_t lds = sizeof(double) * 768 + eigh_lds_bytes(24);
    if (lds > 48 * 1024)
        ET_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fit_finish_eigh_kernel<8, 12>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((fit_finish_eigh_kernel<8, 12>), dim3(2), dim3(kEighThreads), lds, st, sums, k, G_obs, G_pred, count, U_obs,
                       U_pred, sigma_obs, sigma_pred);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

#ifdef ET_EXP_EIGHSTAMP
extern "C" int et_debug_eighstamp(unsigned long long *host, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(et::g_eighstamp), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(et::g_eighstamp), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
#endif
