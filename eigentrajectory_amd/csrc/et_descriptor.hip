// et_descriptor.hip -- projection and (anchor +) reconstruction kernels for gfx950.
//
// Reference semantics (paths relative to the reference repository):
//   projection      EigenTrajectory/descriptor.py:144-160 (+ normalizer.py:17-51)
//   reconstruction  EigenTrajectory/anchor.py:76-88 + descriptor.py:162-176 (+ normalizer.py:53-62)
//   routing         EigenTrajectory/model.py:73-105 (moving / static descriptor per row)
//
// All three kernels are HBM-bound scans (208 / 136 / 136 B per
// trajectory against ~500 flop), so the design goal is full-width coalesced
// traffic: 16-byte-per-lane global accesses on the AoS trajectory rows, staged
// through LDS so that each lane can then own one trajectory (or one
// (trajectory, sample) pair) for the tiny k<=6 contraction.  MFMA is not used there:
// the contraction depth is k=6 / 2T=16..24 and the kernels sit far below the
// VALU roof.  The one exception is the fused best-of-S evaluation epilogue
// (reconstruct_metrics_mfma_kernel, 12 <= S <= 64): 600 B per trajectory against
// 5 760 flop -- its contraction runs on the f16 matrix pipe from two-term f16 splits,
// its inputs travel memory -> LDS without destination registers (round 4).
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "et_common.h"
#include "et_options.h"

namespace et {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int kTile = 256;  // trajectories (or pairs) per workgroup = threads per workgroup

// workgroup -> tile.  Workgroup b runs on XCD b % 8, so the identity map deals consecutive tiles round-robin over the
// XCDs.  The projection instead gives every XCD one contiguous eighth of the tiles: -1.1 ... -2.0 % on its 0.40 ms in
// four same-box A/B runs (profiles/r04d_tile_maps.txt); the S = 1 reconstruction LOSES 3-4 % with the same map and keeps
// the identity.  (-DET_TILE_MAP=0 / 1 forces one map for both; 2: projection walks the rows from the end -- the rows the
// preceding fit read last might still be in the Infinity Cache: they are not, +-0.)
#ifndef ET_TILE_MAP
#define ET_TILE_MAP 3
#endif
template <bool PROJECT = false>
__device__ __forceinline__ int64_t tile_of_block() {
    constexpr bool contiguous = ET_TILE_MAP == 1 || (ET_TILE_MAP == 3 && PROJECT);
    if (ET_TILE_MAP == 2 && PROJECT) return (int64_t)gridDim.x - 1 - blockIdx.x;
    if (contiguous) {
        // XCD x owns the tiles [x q + min(x, r), ...) with q = grid / 8, r = grid % 8: a bijection for EVERY grid size
        const unsigned q = gridDim.x / 8, r = gridDim.x % 8, x = blockIdx.x % 8;
        return (int64_t)(x * q + (x < r ? x : r) + blockIdx.x / 8);
    }
    return (int64_t)blockIdx.x;
}

// ------------------------------------------------------------------------------------------
// Projection, specialised: one workgroup = 256 trajectories.
//   phase 1  coalesced float4 loads of the obs / pred row blocks -> LDS (row pitch padded
//            by 16 B so that the per-lane row reads below are bank-conflict free)
//   phase 2  lane = trajectory: normaliser state, normalise, U^T x (U from LDS), k-major stores
// ------------------------------------------------------------------------------------------
// (round 5: staging the rows memory -> LDS directly, `buffer_load_dwordx4 ... lds`, default and non-temporal policy, measured
// no faster: profiles/r05d_project_dma_ab.txt; source: tools/archive/lost_forms/project_lds_dma.hip.txt)
// Cache policy of the two HBM-bound kernels on BIG launches (round 6, profiles/r06b_project_recon_ab.txt: same-box A/B, five
// alternating rounds per form at N = 1e7).  STREAM = the launch moves more than the 256 MB memory-side cache holds:
//   projection      row loads non-temporal (each row is read once) and C_obs stores non-temporal (nobody re-reads them soon);
//                   C_pred and nrm keep the default policy -- the reconstruction and the k-means read them next.
//                   0.384 -> 0.357 ms (-7 %).  The S = 1 reconstruction behind it pays 0.010 ms of that back (the dirty
//                   C_pred / nrm lines the cache now keeps are written out while it runs): project + reconstruct
//                   0.619 -> 0.602 ms.
//   reconstruction  S > 1 (the model form, 1920 B written per 496 B read): 96-byte row stores non-temporal, tiles walked
//                   from the end: 4.78 -> 4.68 ms at S = 20 (-2 %, four alternating rounds, tools/ab_recon.py).  S = 1: non-temporal row stores cost
//                   the kernel 0.012 ms (and save the farthest-first pass behind it 0.026 ms in the bench's sequence --
//                   not taken: the reconstruction is the stage with a target) -- default policy, identity tile order.
// Small launches (scenes, dataset-sized fits) keep the default policy everywhere: their consumers read the results from
// L2 / the memory-side cache.  Measured and NOT adopted: 128-row projection tiles (6 workgroups per CU instead of 3):
// +2.5 % time; non-temporal C_pred / nrm stores: +1 %.
constexpr int64_t kStreamBytes = 256ll << 20;
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void st_f32(float *p, float v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <bool NT>
__device__ __forceinline__ float4 ld_f32x4(const float4 *p) {
    if constexpr (NT) {
        const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    } else {
        return *p;
    }
}
template <bool NT>
__device__ __forceinline__ void st_f32x4(float4 *p, const float4 &v) {
    if constexpr (NT) __builtin_nontemporal_store((f32x4_nt){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4_nt *>(p));
    else *p = v;
}

// pose (5,N): the normaliser in the form the fused error metric consumes (et_anchor_reconstruct_metrics_pose) -- origin,
// the rotation already multiplied by the scale, and 1 / scale with the moving / static decision in its sign bit:
//   normalise(g) = ((g - o) . (c', s'), (g - o) . (-s', c')),   ||denorm(w) - g|| = ||w - normalise(g)|| * |inv|
// An optional output of the projection (20 B per row): the metric kernel then spends no square root, reciprocal or
// selects on re-deriving them from nrm for every pass.
__device__ __forceinline__ void store_pose(float *__restrict__ pose, int64_t N, int64_t n, const RowNorm &p) {
    pose[n] = p.ox;
    pose[N + n] = p.oy;
    pose[2 * N + n] = p.c * p.sca;
    pose[3 * N + n] = p.s * p.sca;
    pose[4 * N + n] = __uint_as_float(__float_as_uint(p.inv) | (p.mv ? 0x80000000u : 0u));
}

template <int TO, int TP, int K, int TILE, bool STREAM>
__global__ __launch_bounds__(TILE) void project_tile_kernel(
    const float *__restrict__ obs, const float *__restrict__ pred, int64_t N,
    const float *__restrict__ U_obs_m, const float *__restrict__ U_pred_m,
    const float *__restrict__ U_obs_s, const float *__restrict__ U_pred_s,
    int mode, float static_dist,
    float *__restrict__ C_obs, float *__restrict__ C_pred, float *__restrict__ nrm, uint8_t *__restrict__ flag,
    float *__restrict__ pose) {
    constexpr int DO = 2 * TO, DP = 2 * TP;
    constexpr int QO = DO / 4, QP = DP / 4;        // float4 per row
    constexpr int PO = QO + 1, PP = QP + 1;        // padded row pitch in float4 (odd -> conflict free)
    constexpr int UN = (DO + DP) * K;              // floats of U per descriptor
    static_assert(DO % 4 == 0 && DP % 4 == 0, "rows must be float4 multiples");
    static_assert(PO % 2 == 1 && PP % 2 == 1, "padded pitch must be odd in float4 units");

    __shared__ float4 sObs[TILE * PO];
    __shared__ float4 sPred[TILE * PP];
    __shared__ float sU[2 * UN];  // [descriptor: 0 static, 1 moving][obs rows | pred rows][K]

    const int tid = threadIdx.x;
    const int64_t n0 = tile_of_block<true>() * TILE;
    if (n0 >= N) return;
    const int rows = (int)min((int64_t)TILE, N - n0);
    const bool has_pred = pred != nullptr && C_pred != nullptr;

    // ---- phase 1: stage rows (all loads issued before the first LDS write)
    {
        const float4 *g = reinterpret_cast<const float4 *>(obs + n0 * DO);
        float4 v[QO];
#pragma unroll
        for (int j = 0; j < QO; ++j) {
            const int q = tid + j * TILE;
            v[j] = (q < rows * QO) ? ld_f32x4<STREAM>(g + q) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 w[QP];
        if (has_pred) {
            const float4 *gp = reinterpret_cast<const float4 *>(pred + n0 * DP);
#pragma unroll
            for (int j = 0; j < QP; ++j) {
                const int q = tid + j * TILE;
                w[j] = (q < rows * QP) ? ld_f32x4<STREAM>(gp + q) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        for (int i = tid; i < 2 * UN; i += TILE) {
            const int desc = i / UN, r = i - desc * UN;
            const float *src;
            int off;
            if (r < DO * K) {
                src = desc ? U_obs_m : U_obs_s;
                off = r;
            } else {  // the pred half is staged only when it is used: without pred the U_pred operands may belong
                      // to another pred_len (shorter than TP rows) or be absent
                src = has_pred ? (desc ? U_pred_m : U_pred_s) : nullptr;
                off = r - DO * K;
            }
            sU[i] = src ? src[off] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < QO; ++j) {
            const int q = tid + j * TILE;
            sObs[(q / QO) * PO + (q % QO)] = v[j];
        }
        if (has_pred) {
#pragma unroll
            for (int j = 0; j < QP; ++j) {
                const int q = tid + j * TILE;
                sPred[(q / QP) * PP + (q % QP)] = w[j];
            }
        }
    }
    __syncthreads();
    if (tid >= rows) return;

    // ---- phase 2: one lane = one trajectory
    const int64_t n = n0 + tid;
    float xo[DO];
#pragma unroll
    for (int j = 0; j < QO; ++j) {
        const float4 v = sObs[tid * PO + j];
        xo[4 * j] = v.x;
        xo[4 * j + 1] = v.y;
        xo[4 * j + 2] = v.z;
        xo[4 * j + 3] = v.w;
    }
    const float ox = xo[DO - 2], oy = xo[DO - 1];
    const float dx = ox - xo[DO - 6], dy = oy - xo[DO - 5];
    const RowNorm p = row_norm(ox, oy, dx, dy, mode, static_dist);
    if (nrm) {
        st_f32<false>(nrm + n, ox);
        st_f32<false>(nrm + N + n, oy);
        st_f32<false>(nrm + 2 * N + n, dx);
        st_f32<false>(nrm + 3 * N + n, dy);
    }
    if (flag) flag[n] = (uint8_t)p.mv;
    if (pose) store_pose(pose, N, n, p);

    const float *u = sU + p.mv * UN;
    if (C_obs) {
        float acc[K];
#pragma unroll
        for (int j = 0; j < K; ++j) acc[j] = 0.f;
#pragma unroll
        for (int t = 0; t < TO; ++t) {
            float a, b;
            normalize_point(p, xo[2 * t], xo[2 * t + 1], a, b);
#pragma unroll
            for (int j = 0; j < K; ++j) acc[j] = fmaf(u[(2 * t) * K + j], a, acc[j]);
#pragma unroll
            for (int j = 0; j < K; ++j) acc[j] = fmaf(u[(2 * t + 1) * K + j], b, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < K; ++j) st_f32<STREAM>(C_obs + (int64_t)j * N + n, acc[j]);
    }
    if (has_pred) {
        const float *up = u + DO * K;
        float acc[K];
#pragma unroll
        for (int j = 0; j < K; ++j) acc[j] = 0.f;
#pragma unroll
        for (int q = 0; q < QP; ++q) {
            const float4 v = sPred[tid * PP + q];
            float a, b;
            normalize_point(p, v.x, v.y, a, b);
#pragma unroll
            for (int j = 0; j < K; ++j) acc[j] = fmaf(up[(4 * q) * K + j], a, acc[j]);
#pragma unroll
            for (int j = 0; j < K; ++j) acc[j] = fmaf(up[(4 * q + 1) * K + j], b, acc[j]);
            normalize_point(p, v.z, v.w, a, b);
#pragma unroll
            for (int j = 0; j < K; ++j) acc[j] = fmaf(up[(4 * q + 2) * K + j], a, acc[j]);
#pragma unroll
            for (int j = 0; j < K; ++j) acc[j] = fmaf(up[(4 * q + 3) * K + j], b, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < K; ++j) st_f32<false>(C_pred + (int64_t)j * N + n, acc[j]);
    }
}

// Projection, any (T_obs, T_pred, k): one lane = one trajectory, rows read straight
// from global memory.  Correct for every shape the ABI admits; not tuned.
__global__ __launch_bounds__(kTile) void project_generic_kernel(
    const float *__restrict__ obs, const float *__restrict__ pred, int64_t N, int T_obs, int T_pred, int k,
    const float *__restrict__ U_obs_m, const float *__restrict__ U_pred_m,
    const float *__restrict__ U_obs_s, const float *__restrict__ U_pred_s,
    int mode, float static_dist,
    float *__restrict__ C_obs, float *__restrict__ C_pred, float *__restrict__ nrm, uint8_t *__restrict__ flag,
    float *__restrict__ pose) {
    const int64_t n = (int64_t)blockIdx.x * kTile + threadIdx.x;
    if (n >= N) return;
    const float *row = obs + n * 2 * T_obs;
    const float ox = row[2 * (T_obs - 1)], oy = row[2 * (T_obs - 1) + 1];
    const float dx = ox - row[2 * (T_obs - 3)], dy = oy - row[2 * (T_obs - 3) + 1];
    const RowNorm p = row_norm(ox, oy, dx, dy, mode, static_dist);
    if (nrm) {
        nrm[n] = ox;
        nrm[N + n] = oy;
        nrm[2 * N + n] = dx;
        nrm[3 * N + n] = dy;
    }
    if (flag) flag[n] = (uint8_t)p.mv;
    if (pose) store_pose(pose, N, n, p);
    if (C_obs) {
        const float *U = p.mv ? U_obs_m : U_obs_s;
        for (int j = 0; j < k; ++j) {
            float acc = 0.f;
            for (int t = 0; t < T_obs; ++t) {
                float a, b;
                normalize_point(p, row[2 * t], row[2 * t + 1], a, b);
                acc = fmaf(U[(2 * t) * k + j], a, acc);
                acc = fmaf(U[(2 * t + 1) * k + j], b, acc);
            }
            C_obs[(int64_t)j * N + n] = acc;
        }
    }
    if (pred && C_pred) {
        const float *U = p.mv ? U_pred_m : U_pred_s;
        const float *prow = pred + n * 2 * T_pred;
        for (int j = 0; j < k; ++j) {
            float acc = 0.f;
            for (int t = 0; t < T_pred; ++t) {
                float a, b;
                normalize_point(p, prow[2 * t], prow[2 * t + 1], a, b);
                acc = fmaf(U[(2 * t) * k + j], a, acc);
                acc = fmaf(U[(2 * t + 1) * k + j], b, acc);
            }
            C_pred[(int64_t)j * N + n] = acc;
        }
    }
}

// Scene form of the projection (model.py:73-90 for ONE scene batch, obs only): a single workgroup, so that the
// scene-wide mean of the last observed positions is available in the same launch --
//   C_obs (k,N), nrm (4,N), obs_ori (2,N) = obs[:, -1].T - mean over the scene (model.py:86-89), flag (N).
// The reference's real workload (N <= 57 pedestrians per forward) is launch-bound: this is one launch where the
// generic path needs three (projection, mean, subtraction).
__global__ __launch_bounds__(kTile) void scene_project_kernel(
    const float *__restrict__ obs, int N, int T_obs, int k, const float *__restrict__ U_obs_m,
    const float *__restrict__ U_obs_s, int mode, float static_dist, float *__restrict__ C_obs, float *__restrict__ nrm,
    float *__restrict__ obs_ori, uint8_t *__restrict__ flag) {
    __shared__ float sSum[2][kTile / kWave];
    float sx = 0.f, sy = 0.f;
    for (int n = threadIdx.x; n < N; n += kTile) {
        const float *row = obs + (int64_t)n * 2 * T_obs;
        const float ox = row[2 * (T_obs - 1)], oy = row[2 * (T_obs - 1) + 1];
        const float dx = ox - row[2 * (T_obs - 3)], dy = oy - row[2 * (T_obs - 3) + 1];
        const RowNorm p = row_norm(ox, oy, dx, dy, mode, static_dist);
        nrm[n] = ox;
        nrm[N + n] = oy;
        nrm[2 * N + n] = dx;
        nrm[3 * N + n] = dy;
        if (flag) flag[n] = (uint8_t)p.mv;
        sx += ox;
        sy += oy;
        const float *U = p.mv ? U_obs_m : U_obs_s;
        for (int j = 0; j < k; ++j) {
            float acc = 0.f;
            for (int t = 0; t < T_obs; ++t) {
                float a, b;
                normalize_point(p, row[2 * t], row[2 * t + 1], a, b);
                acc = fmaf(U[(2 * t) * k + j], a, acc);
                acc = fmaf(U[(2 * t + 1) * k + j], b, acc);
            }
            C_obs[(int64_t)j * N + n] = acc;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        sx += __shfl_xor(sx, o);
        sy += __shfl_xor(sy, o);
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        sSum[0][threadIdx.x / kWave] = sx;
        sSum[1][threadIdx.x / kWave] = sy;
    }
    __syncthreads();
    float mx = 0.f, my = 0.f;
    for (int w = 0; w < kTile / kWave; ++w) {
        mx += sSum[0][w];
        my += sSum[1][w];
    }
    mx = mx / (float)N;
    my = my / (float)N;
    for (int n = threadIdx.x; n < N; n += kTile) {  // this thread wrote nrm[n], nrm[N + n] itself
        obs_ori[n] = nrm[n] - mx;
        obs_ori[N + n] = nrm[N + n] - my;
    }
}

// Normaliser state of trajectory n from the cached nrm (4,N) or, failing that, from obs.
__device__ __forceinline__ RowNorm load_row_norm(const float *__restrict__ nrm, const float *__restrict__ obs,
                                                 int64_t N, int64_t n, int T_obs, int mode, float static_dist) {
    float ox = 0.f, oy = 0.f, dx = 0.f, dy = 0.f;
    if (mode == ET_MODE_IDENTITY) {
    } else if (nrm) {
        ox = nrm[n];
        oy = nrm[N + n];
        dx = nrm[2 * N + n];
        dy = nrm[3 * N + n];
    } else {
        const float *row = obs + n * 2 * T_obs;
        ox = row[2 * (T_obs - 1)];
        oy = row[2 * (T_obs - 1) + 1];
        dx = ox - row[2 * (T_obs - 3)];
        dy = oy - row[2 * (T_obs - 3) + 1];
    }
    return row_norm(ox, oy, dx, dy, mode, static_dist);
}

constexpr int kNormStride = 8;  // floats per cached RowNorm in LDS
#ifndef ET_RECON_DIRECT
#define ET_RECON_DIRECT 0
#endif
constexpr bool kReconDirect = ET_RECON_DIRECT != 0;  // experiment: per-lane row stores instead of the LDS-staged tile

__device__ __forceinline__ void store_row_norm(float *s, const RowNorm &p) {
    s[0] = p.ox;
    s[1] = p.oy;
    s[2] = p.c;
    s[3] = p.s;
    s[4] = p.sca;
    s[5] = p.inv;
    s[6] = __int_as_float(p.mv);
}

__device__ __forceinline__ RowNorm fetch_row_norm(const float *s) {
    RowNorm p;
    p.ox = s[0];
    p.oy = s[1];
    p.c = s[2];
    p.s = s[3];
    p.sca = s[4];
    p.inv = s[5];
    p.mv = __float_as_int(s[6]);
    return p;
}

// ------------------------------------------------------------------------------------------
// Anchor add + reconstruction, specialised on (T_pred, k); S is a run-time value <= 256.
// One workgroup = TN = 256/S trajectories x S samples; lane = (trajectory, sample) pair,
// numbered nl*S + s so that the k coefficient loads C[j][n][s] are unit-stride across lanes.
// The (S, TN, 2T) result tile is staged in LDS and written back with coalesced float4 stores
// (each sample plane of the (S,N,T,2) output is a contiguous run of TN rows).
// ------------------------------------------------------------------------------------------
#ifndef ET_RECON_TILES
#define ET_RECON_TILES 4
#endif
constexpr int kReconTiles = ET_RECON_TILES;  // consecutive tiles per workgroup when S > 1 (U / anchors staged once)

template <int TP, int K, bool STREAM>
__global__ __launch_bounds__(kTile) void reconstruct_tile_kernel(
    const float *__restrict__ C, int64_t N, int S, int TN, int T_obs,
    const float *__restrict__ obs, const float *__restrict__ nrm,
    const float *__restrict__ A_m, const float *__restrict__ A_s,
    const float *__restrict__ U_m, const float *__restrict__ U_s,
    int mode, float static_dist, float *__restrict__ out) {
    constexpr int DP = 2 * TP, QP = DP / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sOut = smem;                               // rows*S*DP floats (16-B aligned)
    float *sNorm = sOut + TN * S * DP;                // TN * kNormStride
    float *sU = sNorm + TN * kNormStride;             // 2 * DP * K
    float *sA = sU + 2 * DP * K;                      // 2 * K * S

    const int tid = threadIdx.x;
    const int tiles = S == 1 ? 1 : kReconTiles;
    const int nl = tid / S, s = tid - nl * S;
    int64_t n0 = (STREAM ? (int64_t)gridDim.x - 1 - blockIdx.x : tile_of_block()) * tiles * TN;
    if (n0 >= N) return;
    int rows = (int)min((int64_t)TN, N - n0);

    // issue this lane's coefficient loads first: they are in flight while U / anchors / normaliser
    // state are staged (one exposed HBM latency per workgroup instead of two)
    float craw[K];
    if (tid < rows * S) {
#pragma unroll
        for (int j = 0; j < K; ++j) craw[j] = C[((int64_t)j * N + n0 + nl) * S + s];
    }
    RowNorm p;
    if (S == 1) {  // lane == trajectory: the normaliser state never leaves the registers
        if (tid < rows) p = load_row_norm(nrm, obs, N, n0 + nl, T_obs, mode, static_dist);
    } else if (tid < rows) {
        p = load_row_norm(nrm, obs, N, n0 + tid, T_obs, mode, static_dist);
    }
    for (int i = tid; i < 2 * DP * K; i += kTile) {
        const float *src = (i >= DP * K) ? U_m : U_s;
        sU[i] = src ? src[i % (DP * K)] : 0.f;
    }
    for (int i = tid; i < 2 * K * S; i += kTile) {
        const float *src = (i >= K * S) ? A_m : A_s;
        sA[i] = src ? src[i % (K * S)] : 0.f;
    }
    for (int it = 0; it < tiles && rows > 0; ++it) {
        const int npairs = rows * S;
        const int64_t n = n0 + nl;
        if (S != 1 && tid < rows) store_row_norm(sNorm + tid * kNormStride, p);
        __syncthreads();  // U / anchors / this tile's normaliser state staged; the previous tile's write-back is done

        if (tid < npairs) {
            if (S != 1) p = fetch_row_norm(sNorm + nl * kNormStride);
            const float *u = sU + p.mv * DP * K;
            const float *a = sA + p.mv * K * S;
            float c[K];
#pragma unroll
            for (int j = 0; j < K; ++j) c[j] = a[j * S + s] + craw[j];  // anchor.py:87
            float4 *dst = kReconDirect ? reinterpret_cast<float4 *>(out + (((int64_t)s * N + n) * DP))
                                       : reinterpret_cast<float4 *>(sOut + ((size_t)s * rows + nl) * DP);
#pragma unroll
            for (int q = 0; q < QP; ++q) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int f = 4 * q + e;
                    float acc = 0.f;
#pragma unroll
                    for (int j = 0; j < K; ++j) acc = fmaf(u[f * K + j], c[j], acc);  // descriptor.py:87
                    v[e] = acc;
                }
                float4 o;
                denormalize_point(p, v[0], v[1], o.x, o.y);
                denormalize_point(p, v[2], v[3], o.z, o.w);
                dst[q] = o;
            }
        }
        if (kReconDirect) return;
        // the next tile's loads go out before this tile's write-back: they are in flight while the stores drain
        const int64_t n0_next = n0 + TN;
        const int rows_next = (it + 1 < tiles) ? (int)max((int64_t)0, min((int64_t)TN, N - n0_next)) : 0;
        if (tid < rows_next * S) {
#pragma unroll
            for (int j = 0; j < K; ++j) craw[j] = C[((int64_t)j * N + n0_next + nl) * S + s];
        }
        RowNorm p_next;
        if (tid < rows_next) p_next = load_row_norm(nrm, obs, N, n0_next + tid, T_obs, mode, static_dist);
        __syncthreads();

        // coalesced write-back: plane s holds rows*QP consecutive float4 starting at row n0
        const float4 *src4 = reinterpret_cast<const float4 *>(sOut);
        float4 *out4 = reinterpret_cast<float4 *>(out);
        const int per_plane = rows * QP;
        const int total = S * per_plane;
        for (int q = tid; q < total; q += kTile) {
            const int sp = q / per_plane, r = q - sp * per_plane;
            st_f32x4<STREAM>(out4 + ((int64_t)sp * N + n0) * QP + r, src4[q]);
        }
        if (tid < rows_next) p = p_next;
        n0 = n0_next;
        rows = rows_next;
    }
}

// ------------------------------------------------------------------------------------------
// Fused evaluation epilogue: the same anchor-add + reconstruction, but instead of
// writing the (S,N,T,2) trajectories (1920 B per pedestrian at S=20) every pair is compared with the
// ground truth on the spot and only best-of-S ADE / FDE per pedestrian leave the chip (8 B):
//   ADE_n = min_s mean_t ||rec[s,n,t] - gt[n,t]||,  FDE_n = min_s ||rec[s,n,T-1] - gt[n,T-1]||
// (utils/metrics.py:73-102, and the euclidean terms of EigenTrajectory/model.py:120-123).
// ------------------------------------------------------------------------------------------
template <int TP, int K>
__global__ __launch_bounds__(kTile) void reconstruct_metrics_tile_kernel(
    const float *__restrict__ C, int64_t N, int S, int TN, int T_obs,
    const float *__restrict__ obs, const float *__restrict__ nrm,
    const float *__restrict__ A_m, const float *__restrict__ A_s,
    const float *__restrict__ U_m, const float *__restrict__ U_s,
    int mode, float static_dist, const float *__restrict__ gt, float *__restrict__ ade, float *__restrict__ fde) {
    constexpr int DP = 2 * TP, QP = DP / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sGt = smem;                                // TN * DP (16-B aligned)
    float *sMet = sGt + TN * DP;                      // TN * S * 2
    float *sNorm = sMet + TN * S * 2;                 // TN * kNormStride
    float *sU = sNorm + TN * kNormStride;             // 2 * DP * K
    float *sA = sU + 2 * DP * K;                      // 2 * K * S

    const int tid = threadIdx.x;
    const int64_t n0 = (int64_t)blockIdx.x * TN;
    const int rows = (int)min((int64_t)TN, N - n0);
    const int npairs = rows * S;
    const int nl = tid / S, s = tid - nl * S;
    const int64_t n = n0 + nl;
    float craw[K];
    if (tid < npairs) {
        const float *cp = C + (n * S + s);  // + j N S: one 64-bit stride, no per-load index arithmetic
        const int64_t plane = N * S;
#pragma unroll
        for (int j = 0; j < K; ++j) craw[j] = cp[j * plane];
    }
    {
        const float4 *g4 = reinterpret_cast<const float4 *>(gt + n0 * DP);
        float4 *d4 = reinterpret_cast<float4 *>(sGt);
        for (int q = tid; q < rows * QP; q += kTile) d4[q] = g4[q];
    }
    if (tid < rows) store_row_norm(sNorm + tid * kNormStride, load_row_norm(nrm, obs, N, n0 + tid, T_obs, mode, static_dist));
    for (int i = tid; i < 2 * DP * K; i += kTile) {
        const float *src = (i >= DP * K) ? U_m : U_s;
        sU[i] = src ? src[i % (DP * K)] : 0.f;
    }
    for (int i = tid; i < 2 * K * S; i += kTile) {
        const float *src = (i >= K * S) ? A_m : A_s;
        sA[i] = src ? src[i % (K * S)] : 0.f;
    }
    __syncthreads();
    // The displacement norms are taken in the NORMALISED frame: denormalisation is a rotation, a translation and a
    // division by sca, so ||denorm(w) - gt|| = ||w - normalise(gt)|| / sca; the ground truth is normalised once per row
    // (12 points) instead of denormalising 12 points for each of the S samples.
    if (tid < rows) {
        const RowNorm p = fetch_row_norm(sNorm + tid * kNormStride);
        float4 *g4 = reinterpret_cast<float4 *>(sGt + tid * DP);
#pragma unroll
        for (int q = 0; q < QP; ++q) {
            const float4 g = g4[q];
            float4 o;
            normalize_point(p, g.x, g.y, o.x, o.y);
            normalize_point(p, g.z, g.w, o.z, o.w);
            g4[q] = o;
        }
    }
    __syncthreads();

    if (tid < npairs) {
        const RowNorm p = fetch_row_norm(sNorm + nl * kNormStride);
        const float *u = sU + p.mv * DP * K;
        const float *a = sA + p.mv * K * S;
        float c[K];
#pragma unroll
        for (int j = 0; j < K; ++j) c[j] = a[j * S + s] + craw[j];
        const float4 *g4 = reinterpret_cast<const float4 *>(sGt + nl * DP);
        float sum = 0.f, last = 0.f;
#pragma unroll
        for (int q = 0; q < QP; ++q) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int f = 4 * q + e;
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < K; ++j) acc = fmaf(u[f * K + j], c[j], acc);
                v[e] = acc;
            }
            const float4 g = g4[q];  // normalised ground truth
            const float ex = v[0] - g.x, ey = v[1] - g.y, fx = v[2] - g.z, fy = v[3] - g.w;
            // v_sqrt_f32 (1 ulp) instead of the correctly rounded sqrtf() sequence (a dozen VALU instructions each, 24 per
            // pair): the metric is compared at 1e-5 m, the difference is 6e-8 relative
            const float d0 = __builtin_amdgcn_sqrtf(ex * ex + ey * ey), d1 = __builtin_amdgcn_sqrtf(fx * fx + fy * fy);
            sum = (sum + d0) + d1;
            last = d1;
        }
        const float back = p.mv ? p.inv : 1.0f;  // back to metres
        sMet[2 * tid] = (sum / (float)TP) * back;
        sMet[2 * tid + 1] = last * back;
    }
    __syncthreads();
    // best of S per pedestrian: a tree over the S consecutive pairs of a row, every pair's lane taking part
    // (a lane-per-row loop over S left one wavefront busy for 19 dependent steps while the other three waited)
    for (int len = S; len > 1;) {
        const int half = (len + 1) >> 1;
        if (tid < npairs && s + half < len) {  // the partner slot s + half >= len - half is not written in this step
            float2 mine = *reinterpret_cast<const float2 *>(sMet + 2 * tid);
            const float2 other = *reinterpret_cast<const float2 *>(sMet + 2 * (tid + half));
            mine.x = (other.x < mine.x || isnan(other.x)) ? other.x : mine.x;  // torch.min propagates NaN
            mine.y = (other.y < mine.y || isnan(other.y)) ? other.y : mine.y;
            *reinterpret_cast<float2 *>(sMet + 2 * tid) = mine;
        }
        __syncthreads();
        len = half;
    }
    if (tid < rows) {
        ade[n0 + tid] = sMet[2 * (tid * S)];
        fde[n0 + tid] = sMet[2 * (tid * S) + 1];
    }
}

// any-shape fallback: lane = pedestrian, loops over samples and steps
__global__ __launch_bounds__(kTile) void reconstruct_metrics_generic_kernel(
    const float *__restrict__ C, int64_t N, int S, int k, int T_obs, int T_pred,
    const float *__restrict__ obs, const float *__restrict__ nrm,
    const float *__restrict__ A_m, const float *__restrict__ A_s,
    const float *__restrict__ U_m, const float *__restrict__ U_s,
    int mode, float static_dist, const float *__restrict__ gt, float *__restrict__ ade, float *__restrict__ fde) {
    const int64_t n = (int64_t)blockIdx.x * kTile + threadIdx.x;
    if (n >= N) return;
    const RowNorm p = load_row_norm(nrm, obs, N, n, T_obs, mode, static_dist);
    const float *U = p.mv ? U_m : U_s;
    const float *A = p.mv ? A_m : A_s;
    const float *g = gt + n * 2 * T_pred;
    float best_a = 0.f, best_f = 0.f;
    for (int s = 0; s < S; ++s) {
        float c[ET_MAX_K];
        for (int j = 0; j < k; ++j) {
            const float cj = C[((int64_t)j * N + n) * S + s];
            c[j] = A ? A[j * S + s] + cj : cj;
        }
        float sum = 0.f, last = 0.f;
        for (int t = 0; t < T_pred; ++t) {
            float vx = 0.f, vy = 0.f;
            for (int j = 0; j < k; ++j) vx = fmaf(U[(2 * t) * k + j], c[j], vx);
            for (int j = 0; j < k; ++j) vy = fmaf(U[(2 * t + 1) * k + j], c[j], vy);
            float x, y;
            denormalize_point(p, vx, vy, x, y);
            const float ex = x - g[2 * t], ey = y - g[2 * t + 1];
            last = sqrtf(ex * ex + ey * ey);
            sum = sum + last;
        }
        const float va = sum / (float)T_pred;
        if (s == 0 || va < best_a || isnan(va)) best_a = va;
        if (s == 0 || last < best_f || isnan(last)) best_f = last;
    }
    ade[n] = best_a;
    fde[n] = best_f;
}

// Backward of the above w.r.t. C: dtraj (S,N,T,2) -> dC (k,N,S).  Mirror image: coalesced
// float4 loads of the gradient tile into LDS, lane = pair, unit-stride dC stores.
template <int TP, int K>
__global__ __launch_bounds__(kTile) void reconstruct_bwd_tile_kernel(
    const float *__restrict__ dtraj, int64_t N, int S, int TN, int T_obs,
    const float *__restrict__ obs, const float *__restrict__ nrm,
    const float *__restrict__ U_m, const float *__restrict__ U_s,
    int mode, float static_dist, float *__restrict__ dC) {
    constexpr int DP = 2 * TP, QP = DP / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sIn = smem;
    float *sNorm = sIn + TN * S * DP;
    float *sU = sNorm + TN * kNormStride;

    const int tid = threadIdx.x;
    const int64_t n0 = (int64_t)blockIdx.x * TN;
    const int rows = (int)min((int64_t)TN, N - n0);

    const float4 *in4 = reinterpret_cast<const float4 *>(dtraj);
    float4 *dst4 = reinterpret_cast<float4 *>(sIn);
    const int per_plane = rows * QP;
    const int total = S * per_plane;
    {
        // The gradient tile is S runs of per_plane float4, one per sample plane: ALL of a thread's loads (<= QP, as
        // S * TN <= kTile) are issued before the first LDS write -- a load / wait / write loop keeps one 16-byte request
        // per thread in flight and is bound by the memory latency (4.9 TB/s at S = 20), not the bandwidth.  (plane,
        // offset) advance incrementally: one division per thread instead of one per element.
        int sp = tid / per_plane, r = tid - sp * per_plane;
        const int da = kTile / per_plane, db = kTile - da * per_plane;
        float4 buf[QP];
#pragma unroll
        for (int it = 0; it < QP; ++it) {
            const bool in = tid + it * kTile < total;
            buf[it] = in4[in ? ((int64_t)sp * N + n0) * QP + r : (int64_t)0];  // (index 0 is always readable)
            sp += da;
            r += db;
            if (r >= per_plane) {
                r -= per_plane;
                ++sp;
            }
        }
        // unconditional stores (threads past the tile's end hit a spare slot): a conditional store would let the compiler
        // sink each load next to it again
        const int spare = (int)((sU + 2 * DP * K - sIn) / 4);
#pragma unroll
        for (int it = 0; it < QP; ++it) dst4[tid + it * kTile < total ? tid + it * kTile : spare] = buf[it];
    }
    for (int i = tid; i < 2 * DP * K; i += kTile) {
        const float *src = (i >= DP * K) ? U_m : U_s;
        sU[i] = src ? src[i % (DP * K)] : 0.f;
    }
    if (tid < rows) store_row_norm(sNorm + tid * kNormStride, load_row_norm(nrm, obs, N, n0 + tid, T_obs, mode, static_dist));
    __syncthreads();

    const int npairs = rows * S;
    if (tid >= npairs) return;
    const int nl = tid / S, s = tid - nl * S;
    const int64_t n = n0 + nl;
    const RowNorm p = fetch_row_norm(sNorm + nl * kNormStride);
    const float *u = sU + p.mv * DP * K;
    const float4 *src = reinterpret_cast<const float4 *>(sIn + ((size_t)s * rows + nl) * DP);
    float acc[K];
#pragma unroll
    for (int j = 0; j < K; ++j) acc[j] = 0.f;
#pragma unroll
    for (int q = 0; q < QP; ++q) {
        const float4 g = src[q];
        float a, b;
        denormalize_point_bwd(p, g.x, g.y, a, b);
#pragma unroll
        for (int j = 0; j < K; ++j) acc[j] = fmaf(u[(4 * q) * K + j], a, acc[j]);
#pragma unroll
        for (int j = 0; j < K; ++j) acc[j] = fmaf(u[(4 * q + 1) * K + j], b, acc[j]);
        denormalize_point_bwd(p, g.z, g.w, a, b);
#pragma unroll
        for (int j = 0; j < K; ++j) acc[j] = fmaf(u[(4 * q + 2) * K + j], a, acc[j]);
#pragma unroll
        for (int j = 0; j < K; ++j) acc[j] = fmaf(u[(4 * q + 3) * K + j], b, acc[j]);
    }
#pragma unroll
    for (int j = 0; j < K; ++j) dC[((int64_t)j * N + n) * S + s] = acc[j];
}

// ------------------------------------------------------------------------------------------
// Fused evaluation epilogue on the MATRIX cores (T_pred = 12, k = 6, 12 <= S <= 64).  In reconstruct_metrics_tile_kernel
// the contraction U (24 x 6) . (C + A) (6 x S) per trajectory -- 144 fused multiply-adds and as many LDS reads of U per
// (trajectory, sample) pair -- is what the kernel waits for (VALU bound at 0.27 of the HBM roof, S = 20).  U is the same
// for every pair, i.e. this is ONE skinny GEMM V = U . C' with the pairs as columns (rows = the 24 features padded to
// 32, 32 pairs per tile):
//   lane (col, h) of a tile holds rows 8g + 4h + r of column col = the time steps {4g + 2h, 4g + 2h + 1}, x and y adjacent;
//   h = 0 and h = 1 each own 6 of the 12 steps, their displacement sums meet through one v_permlane32_swap.
// Which matrix instruction: v_mfma_f32_32x32x2_f32 reproduces the vector code's fmaf chain over k = 0..5 bit for bit, but
// the fp32 matrix instructions run at the vector rate AND on the vector ALU's multipliers (their busy cycles add to the
// epilogue's, profiles/r04c_metrics_pmc.txt) -- and the vector ALU is what bounds this kernel.  So the default is a
// TWO-TERM f16 SPLIT on the f16 matrix pipe: x = hi + lo, hi = f16(x), lo = f16(x - hi) carries 22 bits of x; the four
// cross products (Uh + Ul)(ch + cl) are exact in the fp32 accumulator; 12 products per lane (its 3 k x 4) fill 6 of the 8
// slots of two v_mfma_f32_32x32x16_f16 that share ONE B operand (bh0 bh1 bh2 bl0 bl1 bl2 . .) against A = (Uh Uh 0 0) and
// (Ul Ul 0 0).  Both operands carry a power-of-two scale (U: 2^10, coefficients + anchors: 2^7; exact, undone by the
// epilogue's fused multiply-subtract) that keeps the low halves out of f16's denormal range.  Against the fp32 chain the
// metrics move by <= 1e-6 relative (tests compare at 2e-6 of the largest value).  A pass whose values leave f16's range
// (|U| >= 32, |coefficient + anchor| >= 256, NaN) takes the fp32 instructions; ET_METRICS_MFMA=f32 forces them.
// Per-row descriptor choice (MODE SPLIT): a tile whose rows all take one descriptor uses that descriptor's A operands;
// a mixed tile sends a column's coefficients to its own descriptor's instructions and zero to the other's.
//
// The workgroup-tile form (and its vector-ALU predecessor) is bound by neither pipe: a workgroup lives through load ->
// barrier -> compute -> barrier -> min tree (five barriers) -> store for ONE tile of 240 pairs.  Hence: a persistent grid
// of AUTONOMOUS wavefronts.  A wavefront takes TNW = 64 / S trajectories (S = 20: 3 trajectories = 60 pairs = two
// 32-column tiles) per pass in its own LDS slice and never meets a workgroup barrier after the prologue.  MODE is a
// template parameter: the pass body is one straight line per mode (the uniform branches on a run-time mode cost ~60
// scalar and ~40 vector instructions per pass).
// ------------------------------------------------------------------------------------------
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
constexpr int kMetWaves = 4;  // wavefronts per workgroup
constexpr int kMetStages = 2; // ring depth = passes in flight per wavefront + 1 (deeper rings cost residency and lose:
                              // 1.97 / 2.13 / 2.34 / 3.04 ms for 2 / 3 / 4 / 6 stages, profiles/r04g_metrics.txt)
constexpr float kMetScaleU = 1024.f, kMetScaleC = 128.f, kMetUnscale = 1.f / (1024.f * 128.f);
constexpr int kMetStage = 9 * 64;  // floats per ring stage: 6 coefficient slots | 2 x 64 ground truth | normaliser state
constexpr int kMetRows = 5;        // >= 64 / S for S >= 12
// floats per wavefront: normalised gt | (ADE, FDE) per pair | 1 / sca (8 slots).  256 floats: with the two ring stages a
// workgroup takes 23 008 B of LDS -- SEVEN workgroups per CU (with kMetRows = 8: 24 160 B, six)
constexpr int kMetSlice = kMetRows * 24 + 2 * 64 + 8;

// memory -> LDS without a destination register: LDS address = M0 + 4 * lane
__device__ __forceinline__ void lds_dma_b32(unsigned __attribute__((ext_vector_type(4))) desc, unsigned lds_addr, unsigned voffset, unsigned soffset) {
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voffset), "s"(desc), "s"(soffset) : "memory");
}

__host__ __device__ constexpr size_t metrics_mfma_lds_floats(int S, int n_desc) {
    return (size_t)kMetWaves * kMetSlice + (((size_t)n_desc * 6 * S + 3) & ~(size_t)3) + (size_t)kMetWaves * kMetStages * kMetStage;
}

#ifdef ET_EXP_METSTAMP  // development aid (tools/archive/metstamp.py): shader cycles (s_memtime) a wavefront spends in the phases of
// a pass, summed over all wavefronts: [0] passes, [1] slice hand-over + stores + requests, [2] wait for this pass's
// inputs, [3] LDS reads + ground-truth normalisation + operands, [4] matrix instructions + hand-over, [5] distances,
// [6] best-of-S, [7] wavefronts
__device__ unsigned long long g_metstamp[8];
#define ET_METSTAMP(i)                                                  \
    do {                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();     \
        ms_acc[i] += t_ - ms_t;                                         \
        ms_t = t_;                                                      \
    } while (0)
#else
#define ET_METSTAMP(i)
#endif

#ifndef ET_EXP_MET
#define ET_EXP_MET 0
#endif
#ifndef ET_MET_WGS
#define ET_MET_WGS 6
#endif
template <int TP, int K, int MODE, bool POSE>
__global__ __launch_bounds__(kMetWaves * 64, MODE == ET_MODE_SPLIT ? 5 : ET_MET_WGS) void reconstruct_metrics_mfma_kernel(
    const float *__restrict__ C, int N, int S, int TNW, int T_obs,
    const float *__restrict__ obs, const float *__restrict__ nrm_or_pose,
    const float *__restrict__ A_m, const float *__restrict__ A_s,
    const float *__restrict__ U_m, const float *__restrict__ U_s,
    float static_dist, const float *__restrict__ gt, float *__restrict__ ade, float *__restrict__ fde, int use_f16) {
    static_assert(TP == 12 && K == 6, "rows = 24 features in a 32-row tile, k = 6 = three k per half");
    static_assert(!POSE || MODE != ET_MODE_IDENTITY, "the identity mode has no normaliser to precompute");
    // POSE: the normaliser arrives as pose (5,N) = ox, oy, c sca, s sca, +-1 / sca (store_pose) instead of nrm (4,N)
    const float *nrm = nrm_or_pose;
    constexpr int kPlanes = POSE ? 5 : 4;
    constexpr int DP = 2 * TP, D = kMetStages;
    constexpr bool SPLIT = MODE == ET_MODE_SPLIT;
    constexpr int ND = SPLIT ? 2 : 1;  // descriptors in play: SPLIT 0 static / 1 moving, else the mode's own
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *sGn = smem + wave * kMetSlice;  // TNW * DP (16-B aligned rows)
    float *sMet = sGn + kMetRows * DP;     // 2 * 64
    float *sBack = sMet + 2 * 64;          // TNW
    float *sA = smem + kMetWaves * kMetSlice;  // ND * K * S anchors [descriptor][k][s] * 2^7, shared by the workgroup
    float *sRing = sA + ((ND * K * S + 3) & ~3) + wave * (D * kMetStage);
    const int col_in_tile = lane & 31, h = lane >> 5;

    // A operands of lane (f, h): U[phi(f)][2 j + h] * 2^10, j = 0..2 (f = the lane's tile row; rows 24..31 are padding).
    // Tile row 4 m + i holds feature phi = 4 m + (x_A, x_B, y_A, y_B)[i] of the step pair (A, B) = (2 m, 2 m + 1) -- rows 1
    // and 2 of every group of four swapped against the feature order (x_A, y_A, x_B, y_B) --, so a lane's accumulator pairs are
    // (x_A, x_B) and (y_A, y_B): the epilogue squares and adds two steps per packed instruction.
    const int feat_of_row = (col_in_tile & ~3) + ((col_in_tile & 1) << 1) + ((col_in_tile >> 1) & 1);
    float aU[ND][3];
    f16x8_t aHi[ND], aLo[ND];  // (Uh0 Uh1 Uh2 Uh0 Uh1 Uh2 0 0), (Ul0 Ul1 Ul2 Ul0 Ul1 Ul2 0 0)
    bool u_small = true;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const float *U = (SPLIT ? d == 1 : MODE == ET_MODE_MOVING) ? U_m : U_s;
        _Float16 uh[3], ul[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float u = (U && col_in_tile < DP) ? U[feat_of_row * K + 2 * j + h] * kMetScaleU : 0.f;
            aU[d][j] = u;
            u_small = u_small && fabsf(u) < 32768.f;
            uh[j] = (_Float16)u;
            ul[j] = (_Float16)(u - (float)uh[j]);
        }
        const _Float16 z = (_Float16)0.f;
        aHi[d] = f16x8_t{uh[0], uh[1], uh[2], uh[0], uh[1], uh[2], z, z};
        aLo[d] = f16x8_t{ul[0], ul[1], ul[2], ul[0], ul[1], ul[2], z, z};
    }
    const bool f16_ok = use_f16 && __ballot(!u_small) == 0ull;
    for (int i = tid; i < ND * K * S; i += kMetWaves * 64) {
        const float *src = (SPLIT ? i >= K * S : MODE == ET_MODE_MOVING) ? A_m : A_s;
        sA[i] = src ? src[i % (K * S)] * kMetScaleC : 0.f;
    }
    __syncthreads();  // the only workgroup barrier

    auto wave_sync = [&]() {  // LDS hand-over between the lanes of this wavefront
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    const int64_t plane = (int64_t)N * S;
    const int n_pass = (N + TNW - 1) / TNW;
    const int stride = (int)gridDim.x * kMetWaves;
    const int full_pairs = TNW * S;  // > 32 for every 12 <= S <= 64: both tiles are always in use
    // Everything a lane derives from its position alone is computed ONCE (the integer divisions by S, the anchor and
    // ground-truth offsets): the loop was bound by instruction issue (~1000 instructions per pass, most of them index
    // arithmetic) before this.
    int row_t[2];
    const float *an_t[2];  // tile t: this lane's row of the pass, and its anchor column &sA[(0 * 2 + h) * S + sample]
    bool col_ok[2];
    unsigned long long tile_rows[2];  // (SPLIT) bit 12 r: row r has columns in tile t
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int col = 32 * t + col_in_tile;
        col_ok[t] = col < full_pairs;
        row_t[t] = col_ok[t] ? col / S : 0;
        an_t[t] = sA + h * S + (col_ok[t] ? col - row_t[t] * S : 0);
        const int r_lo = (32 * t) / S, r_hi = min(TNW - 1, (32 * t + 31) / S);
        unsigned long long m = 0;
        for (int r = r_lo; r <= r_hi; ++r) m |= 1ull << (r * TP);
        tile_rows[t] = m;
    }
    // ground-truth point of this lane: (row gr, step gs) if lane < TNW * 12
    const int gr = lane / TP, gs = lane - gr * TP;
    const bool g_ok = gr < TNW;
    // best-of-S: four lanes per row, each takes every fourth sample
    const int mr = lane >> 2, mq = lane & 3;
    const int mrow = mr < TNW ? mr : TNW - 1;

    // A pass's inputs travel from memory STRAIGHT INTO LDS (buffer_load ... lds: no destination registers), D - 1 passes
    // ahead, into a ring of D stages per wavefront.  The loads are inline assembly on purpose: for the compiler's own
    // LDS-DMA intrinsic the wait-count pass puts s_waitcnt vmcnt(0) in front of EVERY later LDS read of the kernel (it
    // cannot tell the stages apart).  Here the waits are ours: every iteration issues exactly 2 stores + 9 loads (the tail
    // re-requests the last pass), so "stage i has landed" is vmcnt <= 11 (D - 1) -- a constant.  Every pass is a FULL pass:
    // the last one is moved back to end at row N (it recomputes a few rows of its predecessor -- same values, same
    // addresses), so the loop has no load or store under a branch (the compiler answers those with full drains).
    // Descriptors live in scalar registers, the per-lane byte offsets in one vector register each for the whole loop, the
    // pass's position is a scalar offset (offsets must fit 32 bits: the host takes this kernel for 2 N S 4 B < 2^32 only).
    auto desc_of = [](const void *base, int64_t bytes) {
        const unsigned long long b = reinterpret_cast<unsigned long long>(base);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
        const unsigned nb = __builtin_amdgcn_readfirstlane((unsigned)(bytes > 0xffffffffll ? 0xffffffffll : bytes));
        return u32x4_t{lo, hi & 0xffffu, nb, 0x00020000u};
    };
    auto rsrc_of = [](const void *base, int64_t bytes) {
        const unsigned long long b = reinterpret_cast<unsigned long long>(base);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
        const int nb = __builtin_amdgcn_readfirstlane((int)(bytes > 0xffffffffll ? 0xffffffffll : bytes));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0, nb, 0x00020000);
    };
    const u32x4_t dc0 = desc_of(C, 2 * plane * 4), dc1 = desc_of(C + 2 * plane, 2 * plane * 4), dc2 = desc_of(C + 4 * plane, 2 * plane * 4);
    const u32x4_t dg = desc_of(gt, (int64_t)N * DP * 4), dn = nrm ? desc_of(nrm, (int64_t)kPlanes * N * 4) : dg;  // (without nrm: a readable dummy)
    const __amdgpu_buffer_rsrc_t ra = rsrc_of(ade, (int64_t)N * 4), rf = rsrc_of(fde, (int64_t)N * 4);
    unsigned oc[2];  // tile t's coefficient loads: (h * plane + column) * 4
#pragma unroll
    for (int t = 0; t < 2; ++t) oc[t] = 4u * (unsigned)((int64_t)h * plane + (col_ok[t] ? 32 * t + col_in_tile : 0));
    // ground truth: the pass's TNW rows are TNW * 24 contiguous floats -> two loads, lane = float index
    const unsigned og0 = 4u * (unsigned)(lane < TNW * DP ? lane : 0), og1 = 4u * (unsigned)(64 + lane < TNW * DP ? 64 + lane : 0);
    // normaliser state [4][N]: lane j * TNW + r loads plane j, row r
    const unsigned onr = (nrm && lane < kPlanes * TNW) ? 4u * (unsigned)((int64_t)(lane / TNW) * N + lane % TNW) : 0u;
    const bool use_nrm = nrm != nullptr && MODE != ET_MODE_IDENTITY;
    const unsigned ring_addr = __builtin_amdgcn_readfirstlane(
        (unsigned)reinterpret_cast<unsigned long long>((__attribute__((address_space(3))) float *)sRing));
    auto first_row = [&](int ps) { return min(ps * TNW, N - TNW); };
    auto issue = [&](int ps, int stage) {  // (passes beyond the last re-request the last one: same count every time)
#if ET_EXP_MET == 3  // measurement aid: every pass re-requests pass 0's inputs (cache resident: no memory traffic)
        ps = 0;
#endif
        const int n0 = __builtin_amdgcn_readfirstlane(first_row(min(ps, n_pass - 1)));
        const unsigned base = ring_addr + (unsigned)stage * (unsigned)(kMetStage * 4);
        const unsigned so_c = (unsigned)n0 * (unsigned)(S * 4), so_g = (unsigned)n0 * (unsigned)(DP * 4), so_n = (unsigned)n0 * 4u;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            lds_dma_b32(dc0, base + (3 * t + 0) * 256, oc[t], so_c);
            lds_dma_b32(dc1, base + (3 * t + 1) * 256, oc[t], so_c);
            lds_dma_b32(dc2, base + (3 * t + 2) * 256, oc[t], so_c);
        }
        lds_dma_b32(dg, base + 6 * 256, og0, so_g);
        lds_dma_b32(dg, base + 7 * 256, og1, so_g);
        lds_dma_b32(dn, base + 8 * 256, onr, so_n);
    };
    int pass = (int)blockIdx.x * kMetWaves + wave;
    const bool did_any = pass < n_pass;
    if (did_any) {
#pragma unroll
        for (int d = 0; d < D - 1; ++d) issue(pass + d * stride, d);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // A pass's two results are stored at the START of the next pass, in front of that pass's loads, by ALL lanes,
    // unconditionally, through a scalar base + a per-lane offset that lives in one register for the whole loop: lanes
    // beyond the pass's rows repeat its last row (same value, same address).  The first iteration has nothing to store
    // yet: it writes zeros to its own pass's slots, which the second iteration overwrites.
    const int om = 4 * mrow;
    auto store_held = [&](float2 v, int n0) {
        const int so = __builtin_amdgcn_readfirstlane(n0 * 4);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.x), ra, om, so, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.y), rf, om, so, 0);
    };
    auto mfma16 = [](f16x8_t hi, f16x8_t lo, u32x4_t bq) {
        const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const f16x8_t bv = __builtin_bit_cast(f16x8_t, bq);
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(lo, bv, __builtin_amdgcn_mfma_f32_32x32x16_f16(hi, bv, zero, 0, 0, 0), 0, 0, 0);
    };
    float2 held = make_float2(0.f, 0.f);
    int held_n0 = did_any ? first_row(pass) : 0;
    // best-of-S: this lane's first (ADE, FDE) pair in the slice
    const u32x2_t *m2 = reinterpret_cast<const u32x2_t *>(sMet) + mrow * S + mq;
    const int nq = S / 4;  // samples mq, mq + 4, ..., mq + 4 (nq - 1) exist for every lane; one more for mq < S % 4
    // (Instantiating the pass body once per ring stage -- every LDS address a loop-invariant register + an immediate
    // offset -- saves ~10 vector instructions per pass and costs 4 registers, i.e. the sixth wavefront per SIMD: +-0.)
    int stage = 0;  // ring slot of the current pass
#ifdef ET_EXP_METSTAMP
    unsigned long long ms_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ms_t = __builtin_amdgcn_s_memtime();
#endif
    for (; pass < n_pass; pass += stride) {
        const int n0 = first_row(pass);
#ifdef ET_EXP_METSTAMP
        ms_t = __builtin_amdgcn_s_memtime();
        ms_acc[0] += 1;
#endif
        wave_sync();  // the previous pass is done with the slice
        store_held(held, held_n0);
        issue(pass + (D - 1) * stride, stage == 0 ? D - 1 : stage - 1);  // travels while D - 1 passes are computed
        ET_METSTAMP(1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(11 * (D - 1)) : "memory");
        ET_METSTAMP(2);
        const float *sIn = sRing + stage * kMetStage;
        stage = stage + 1 == D ? 0 : stage + 1;
        float cur[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 3; ++j) cur[t][j] = sIn[(3 * t + j) * 64 + lane];
        int mv_lane = 0;
        if (g_ok) {
            const float2 gp = *reinterpret_cast<const float2 *>(sIn + 6 * 64 + gr * DP + 2 * gs);
            float2 o;  // (stored NEGATED: the epilogue's fused multiply-add then takes it as it is)
            float back;
            if constexpr (POSE) {
                // normalizer.py:42-51 with the rotation and the scale in one pair: ((p - ori) @ R) * sca = (t . (c', s'), t . (-s', c'))
                const float ox = sIn[8 * 64 + gr], oy = sIn[8 * 64 + TNW + gr], cs = sIn[8 * 64 + 2 * TNW + gr], ss = sIn[8 * 64 + 3 * TNW + gr];
                const float bk = sIn[8 * 64 + 4 * TNW + gr];
                mv_lane = (int)(__float_as_uint(bk) >> 31);  // (SPLIT: the row takes the moving descriptor)
                back = fabsf(bk);
                const float tx = gp.x - ox, ty = gp.y - oy;
                o.x = -(tx * cs + ty * ss);
                o.y = tx * ss - ty * cs;
            } else {
                float ox = sIn[8 * 64 + gr], oy = sIn[8 * 64 + TNW + gr], dx = sIn[8 * 64 + 2 * TNW + gr], dy = sIn[8 * 64 + 3 * TNW + gr];
                if (!use_nrm && MODE != ET_MODE_IDENTITY) {  // (no cached state: from the observed row)
                    const float *row = obs + (int64_t)(n0 + gr) * 2 * T_obs;
                    ox = row[2 * (T_obs - 1)];
                    oy = row[2 * (T_obs - 1) + 1];
                    dx = ox - row[2 * (T_obs - 3)];
                    dy = oy - row[2 * (T_obs - 3) + 1];
                }
                // normaliser state as row_norm() forms it, with the hardware's 1-ulp square root and reciprocal in place of
                // the correctly rounded sequences (~40 instructions per pass on every lane): the metric is compared at
                // 1e-5 m, these differ by 1e-7 relative.  The moving / static decision (model.py:46) keeps row_norm's exact test.
                int mv = MODE == ET_MODE_MOVING;
                if (SPLIT) {
                    const float hx = dx * 0.5f, hy = dy * 0.5f;
                    mv = sqrtf(hx * hx + hy * hy) > static_dist ? 1 : 0;
                }
                mv_lane = mv;
                float c = 1.f, sn = 0.f, sca = 1.f;
                back = 1.f;
                if (MODE != ET_MODE_IDENTITY) {
                    const float r2 = dx * dx + dy * dy;
                    const float r = __builtin_amdgcn_sqrtf(r2), ir = __builtin_amdgcn_rcpf(r);
                    const bool still = !(r > 0.0f);
                    c = still ? (isnan(r) ? r : 1.0f) : dx * ir;
                    sn = still ? (isnan(r) ? r : 0.0f) : dy * ir;
                    sca = mv ? ir * 2.0f : 1.0f;
                    back = mv ? r * 0.5f : 1.0f;
                } else {
                    ox = 0.f;
                    oy = 0.f;
                }
                // normalizer.py:42-51: ((p - ori) @ R) * sca;  ||denorm(w) - gt|| = ||w - normalise(gt)|| / sca
                const float tx = gp.x - ox, ty = gp.y - oy;
                o.x = (tx * c + ty * sn) * -sca;
                o.y = (tx * (-sn) + ty * c) * -sca;
            }
            float *og = sGn + gr * DP + 4 * (gs >> 1) + (gs & 1);  // (x_A, x_B, y_A, y_B) per step pair, like the tile rows
            og[0] = o.x;
            og[2] = o.y;
            if (gs == 0) sBack[gr] = back;
        }
        // bit 12 r (+ step): row r of the pass takes the moving descriptor.  (Every lane asks: a ballot inside the branch
        // above would be seen by the lanes that took it only.)
        const unsigned long long mvbits = SPLIT ? __ballot(mv_lane != 0) : 0ull;
        // Both tiles of the pass in ONE straight line: their LDS round trips, matrix instructions and square roots overlap
        float b[2][3];  // (coefficient + anchor) * 2^7 (anchor.py:87; padding columns: never stored)
        int mvt[2] = {0, 0};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float *an = an_t[t];
            if (SPLIT) {
                mvt[t] = (int)((mvbits >> (row_t[t] * TP)) & 1ull);
                an += mvt[t] * (K * S);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) b[t][j] = fmaf(cur[t][j], kMetScaleC, an[2 * j * S]);
        }
        const float big = fmaxf(fmaxf(fmaxf(fabsf(b[0][0]), fabsf(b[0][1])), fabsf(b[0][2])),
                                fmaxf(fmaxf(fabsf(b[1][0]), fabsf(b[1][1])), fabsf(b[1][2])));
        f32x16_t acc[2];
        ET_METSTAMP(3);
#if ET_EXP_MET == 2  // measurement aid: no operand split, no matrix instructions
        if (1) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = b[t][i % 3] * aU[0][i % 3] + big;
        } else
#endif
        if (f16_ok && __ballot(!(big < 32768.f)) == 0ull) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f16x2_t p0 = __builtin_convertvector((f32x2_t){b[t][0], b[t][1]}, f16x2_t);
                const float r0 = b[t][0] - (float)p0.x, r1 = b[t][1] - (float)p0.y;
                const f16x2_t p1 = __builtin_convertvector((f32x2_t){b[t][2], r0}, f16x2_t);
                const float r2 = b[t][2] - (float)p1.x;
                const f16x2_t p2 = __builtin_convertvector((f32x2_t){r1, r2}, f16x2_t);
                const unsigned q0 = __builtin_bit_cast(unsigned, p0), q1 = __builtin_bit_cast(unsigned, p1), q2 = __builtin_bit_cast(unsigned, p2);
                // (the fourth register meets the zero slots of A: any finite pattern does)
                if (!SPLIT) {
                    acc[t] = mfma16(aHi[0], aLo[0], (u32x4_t){q0, q1, q2, q0});
                } else {
                    const bool any_m = (mvbits & tile_rows[t]) != 0ull, any_s = (~mvbits & tile_rows[t]) != 0ull;  // (scalar)
                    if (!any_m) {
                        acc[t] = mfma16(aHi[0], aLo[0], (u32x4_t){q0, q1, q2, q0});
                    } else if (!any_s) {
                        acc[t] = mfma16(aHi[ND - 1], aLo[ND - 1], (u32x4_t){q0, q1, q2, q0});
                    } else {
                        const unsigned s0 = mvt[t] ? 0u : q0, s1 = mvt[t] ? 0u : q1, s2 = mvt[t] ? 0u : q2;
                        const unsigned m0 = mvt[t] ? q0 : 0u, m1 = mvt[t] ? q1 : 0u, m2 = mvt[t] ? q2 : 0u;
                        const f16x8_t bm = __builtin_bit_cast(f16x8_t, (u32x4_t){m0, m1, m2, m0});
                        f32x16_t a = mfma16(aHi[0], aLo[0], (u32x4_t){s0, s1, s2, s0});
                        a = __builtin_amdgcn_mfma_f32_32x32x16_f16(aHi[ND - 1], bm, a, 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aLo[ND - 1], bm, a, 0, 0, 0);
                    }
                }
            }
        } else {
            // fp32 matrix instructions: the vector code's fmaf chain over k = 0..5, bit for bit (the scales are exact)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16_t a = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 3; ++j) a = __builtin_amdgcn_mfma_f32_32x32x2f32(aU[0][j], (SPLIT && mvt[t]) ? 0.f : b[t][j], a, 0, 0, 0);
                if (SPLIT) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) a = __builtin_amdgcn_mfma_f32_32x32x2f32(aU[ND - 1][j], mvt[t] ? b[t][j] : 0.f, a, 0, 0, 0);
                }
                acc[t] = a;
            }
        }
        wave_sync();  // the normalised ground truth is in the slice
        ET_METSTAMP(4);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            // rows 8 g + 4 h + (0..3) of this column: steps 4 g + 2 h and 4 g + 2 h + 1 (g = 3 is padding)
            const float4 *g4 = reinterpret_cast<const float4 *>(sGn + row_t[t] * DP + 4 * h);
            const float back = sBack[row_t[t]];
            const f32x2_t un = {kMetUnscale, kMetUnscale};
            f32x2_t sum2 = {0.f, 0.f};
            float last = 0.f;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const float4 gn = g4[2 * g];  // = -(normalised ground truth): (x_A, x_B, y_A, y_B)
                // (two steps per instruction: v_pk_fma_f32 / v_pk_mul_f32 on adjacent registers)
                const f32x2_t ex = __builtin_elementwise_fma((f32x2_t){acc[t][4 * g], acc[t][4 * g + 1]}, un, (f32x2_t){gn.x, gn.y});
                const f32x2_t ey = __builtin_elementwise_fma((f32x2_t){acc[t][4 * g + 2], acc[t][4 * g + 3]}, un, (f32x2_t){gn.z, gn.w});
                const f32x2_t d2 = __builtin_elementwise_fma(ey, ey, ex * ex);
                // v_sqrt_f32 (1 ulp): the metric is compared at 1e-5 m
#if ET_EXP_MET == 1  // measurement aid: no square roots
                const f32x2_t d = d2;
#else
                const f32x2_t d = {__builtin_amdgcn_sqrtf(d2.x), __builtin_amdgcn_sqrtf(d2.y)};
#endif
                sum2 = sum2 + d;
                last = d.y;  // h = 1, g = 2: step 11
            }
            const float sum = sum2.x + sum2.y;
            // the other half's six steps: v_permlane32_swap (x, x) leaves x[lane + 32] in the second result's low half
            const u32x2_t ws = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
            const u32x2_t wl = __builtin_amdgcn_permlane32_swap(__float_as_uint(last), __float_as_uint(last), false, false);
            const float other_sum = __uint_as_float(ws.y), other_last = __uint_as_float(wl.y);
            if (h == 0 && col_ok[t])
                *reinterpret_cast<float2 *>(sMet + 2 * (32 * t + col_in_tile)) =
                    make_float2(((sum + other_sum) * (1.0f / (float)TP)) * back, other_last * back);
        }
        wave_sync();
        ET_METSTAMP(5);
        {   // best of S (torch.min propagates NaN): lane (mr, mq) takes samples mq, mq + 4, ...; then the quad's four meet.
            // The metrics are >= +0 or NaN: as unsigned integers their order is the floats' and every NaN is above +inf,
            // so min AND max of the bit patterns decide -- max > 0x7f800000 means a NaN was among them.
            u32x2_t mn = m2[0], mx = mn;
            int k = 1;
            for (; k + 1 < nq; k += 2) {  // (a scalar trip count: two samples per step)
                const u32x2_t v0 = m2[4 * k], v1 = m2[4 * k + 4];
                mn.x = min(min(mn.x, v0.x), v1.x);
                mn.y = min(min(mn.y, v0.y), v1.y);
                mx.x = max(max(mx.x, v0.x), v1.x);
                mx.y = max(max(mx.y, v0.y), v1.y);
            }
            if (k < nq || (S & 3)) {
                const u32x2_t v0 = m2[k < nq ? 4 * k : 0], v1 = m2[mq + 4 * nq < S ? 4 * nq : 0];  // (absent: the first again)
                mn.x = min(min(mn.x, v0.x), v1.x);
                mn.y = min(min(mn.y, v0.y), v1.y);
                mx.x = max(max(mx.x, v0.x), v1.x);
                mx.y = max(max(mx.y, v0.y), v1.y);
            }
#define ET_QUAD(v, ctrl) (unsigned)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, 0xf, 0xf, true)
            mn.x = min(mn.x, ET_QUAD(mn.x, 0xB1)); mn.y = min(mn.y, ET_QUAD(mn.y, 0xB1));  // lane ^ 1
            mx.x = max(mx.x, ET_QUAD(mx.x, 0xB1)); mx.y = max(mx.y, ET_QUAD(mx.y, 0xB1));
            mn.x = min(mn.x, ET_QUAD(mn.x, 0x4E)); mn.y = min(mn.y, ET_QUAD(mn.y, 0x4E));  // lane ^ 2
            mx.x = max(mx.x, ET_QUAD(mx.x, 0x4E)); mx.y = max(mx.y, ET_QUAD(mx.y, 0x4E));
#undef ET_QUAD
            held = make_float2(__uint_as_float(mx.x > 0x7f800000u ? mx.x : mn.x), __uint_as_float(mx.y > 0x7f800000u ? mx.y : mn.y));
            held_n0 = n0;
        }
#ifdef ET_EXP_METSTAMP
        asm volatile("s_nop 0" ::"v"(held.x), "v"(held.y));
#endif
        ET_METSTAMP(6);
    }
#ifdef ET_EXP_METSTAMP
    if (lane == 0) {
        for (int i = 0; i < 7; ++i) atomicAdd(&g_metstamp[i], ms_acc[i]);
        atomicAdd(&g_metstamp[7], 1ull);
    }
#endif
    if (did_any) store_held(held, held_n0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the tail's surplus requests still write into this wavefront's LDS)
}

// Any-shape fallbacks: lane = (trajectory, sample) pair, direct global accesses.
__global__ __launch_bounds__(kTile) void reconstruct_generic_kernel(
    const float *__restrict__ C, int64_t N, int S, int k, int T_obs, int T_pred,
    const float *__restrict__ obs, const float *__restrict__ nrm,
    const float *__restrict__ A_m, const float *__restrict__ A_s,
    const float *__restrict__ U_m, const float *__restrict__ U_s,
    int mode, float static_dist, float *__restrict__ out) {
    const int64_t pair = (int64_t)blockIdx.x * kTile + threadIdx.x;
    if (pair >= N * S) return;
    const int64_t n = pair / S;
    const int s = (int)(pair - n * S);
    const RowNorm p = load_row_norm(nrm, obs, N, n, T_obs, mode, static_dist);
    const float *U = p.mv ? U_m : U_s;
    const float *A = p.mv ? A_m : A_s;
    float c[ET_MAX_K];
    for (int j = 0; j < k; ++j) {
        const float cj = C[((int64_t)j * N + n) * S + s];
        c[j] = A ? A[j * S + s] + cj : cj;
    }
    float *dst = out + (((int64_t)s * N + n) * T_pred) * 2;
    for (int t = 0; t < T_pred; ++t) {
        float vx = 0.f, vy = 0.f;
        for (int j = 0; j < k; ++j) vx = fmaf(U[(2 * t) * k + j], c[j], vx);
        for (int j = 0; j < k; ++j) vy = fmaf(U[(2 * t + 1) * k + j], c[j], vy);
        float x, y;
        denormalize_point(p, vx, vy, x, y);
        dst[2 * t] = x;
        dst[2 * t + 1] = y;
    }
}

__global__ __launch_bounds__(kTile) void reconstruct_bwd_generic_kernel(
    const float *__restrict__ dtraj, int64_t N, int S, int k, int T_obs, int T_pred,
    const float *__restrict__ obs, const float *__restrict__ nrm,
    const float *__restrict__ U_m, const float *__restrict__ U_s,
    int mode, float static_dist, float *__restrict__ dC) {
    const int64_t pair = (int64_t)blockIdx.x * kTile + threadIdx.x;
    if (pair >= N * S) return;
    const int64_t n = pair / S;
    const int s = (int)(pair - n * S);
    const RowNorm p = load_row_norm(nrm, obs, N, n, T_obs, mode, static_dist);
    const float *U = p.mv ? U_m : U_s;
    const float *g = dtraj + (((int64_t)s * N + n) * T_pred) * 2;
    for (int j = 0; j < k; ++j) {
        float acc = 0.f;
        for (int t = 0; t < T_pred; ++t) {
            float a, b;
            denormalize_point_bwd(p, g[2 * t], g[2 * t + 1], a, b);
            acc = fmaf(U[(2 * t) * k + j], a, acc);
            acc = fmaf(U[(2 * t + 1) * k + j], b, acc);
        }
        dC[((int64_t)j * N + n) * S + s] = acc;
    }
}

static bool need_m(int mode) { return mode == ET_MODE_MOVING || mode == ET_MODE_SPLIT; }
static bool need_s(int mode) { return mode != ET_MODE_MOVING; }

// (persistent, barrier-free STREAMING forms of the one-descriptor projection / S = 1 reconstruction were built in round 4 and lost
// to the workgroup-tile kernels -- project 0.414 against 0.386 ms, reconstruct 0.253 against 0.228 ms at N = 1e7, same box,
// profiles/r04a_stream_ab.txt; their source: tools/archive/lost_forms/project_reconstruct_stream.hip.txt)
static int cu_count() {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
        n = 256;
    return n;
}

static bool dims_ok(int T_obs, int T_pred, int k) {
    return T_obs >= 3 && T_obs <= ET_MAX_T && T_pred >= 1 && T_pred <= ET_MAX_T && k >= 1 && k <= ET_MAX_K;
}

}  // namespace et

using namespace et;

extern "C" int et_norm_project(const float *obs, const float *pred, int64_t N, int T_obs, int T_pred, int k,
                               const float *U_obs_m, const float *U_pred_m, const float *U_obs_s,
                               const float *U_pred_s, int mode, float static_dist, float *C_obs, float *C_pred,
                               float *nrm, uint8_t *flag, et_stream_t stream) {
    return et_norm_project_pose(obs, pred, N, T_obs, T_pred, k, U_obs_m, U_pred_m, U_obs_s, U_pred_s, mode, static_dist, C_obs,
                                C_pred, nrm, flag, nullptr, stream);
}

extern "C" int et_norm_project_pose(const float *obs, const float *pred, int64_t N, int T_obs, int T_pred, int k,
                                    const float *U_obs_m, const float *U_pred_m, const float *U_obs_s,
                                    const float *U_pred_s, int mode, float static_dist, float *C_obs, float *C_pred,
                                    float *nrm, uint8_t *flag, float *pose, et_stream_t stream) {
    if (N < 0 || !dims_ok(T_obs, T_pred, k) || mode < 0 || mode > 3) return ET_ERR_INVALID_ARG;
    if (N == 0) return ET_OK;
    if (!obs) return ET_ERR_INVALID_ARG;
    if (C_obs && ((need_m(mode) && !U_obs_m) || (need_s(mode) && !U_obs_s))) return ET_ERR_INVALID_ARG;
    if (pred && C_pred && ((need_m(mode) && !U_pred_m) || (need_s(mode) && !U_pred_s))) return ET_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)ceil_div(N, kTile);
    const bool fast = T_obs == 8 && (!pred || T_pred == 12) && k == 6 && aligned16(obs) && (!pred || aligned16(pred));
    if (fast) {
        const bool stream = N * (int64_t)(pred ? 224 : 104) > kStreamBytes;
        auto kern = stream ? project_tile_kernel<8, 12, 6, kTile, true> : project_tile_kernel<8, 12, 6, kTile, false>;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kTile), 0, st, obs, pred, N, U_obs_m, U_pred_m, U_obs_s, U_pred_s, mode,
                           static_dist, C_obs, C_pred, nrm, flag, pose);
    } else {
        hipLaunchKernelGGL(project_generic_kernel, dim3(grid), dim3(kTile), 0, st, obs, pred, N, T_obs, T_pred, k,
                           U_obs_m, U_pred_m, U_obs_s, U_pred_s, mode, static_dist, C_obs, C_pred, nrm, flag, pose);
    }
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_scene_project(const float *obs, int64_t N, int T_obs, int k, const float *U_obs_m,
                                const float *U_obs_s, int mode, float static_dist, float *C_obs, float *nrm,
                                float *obs_ori, uint8_t *flag, et_stream_t stream) {
    if (N < 0 || N > ET_SCENE_MAX_N || !dims_ok(T_obs, 1, k) || mode < 0 || mode > 3) return ET_ERR_INVALID_ARG;
    if (N == 0) return ET_OK;
    if (!obs || !C_obs || !nrm || !obs_ori) return ET_ERR_INVALID_ARG;
    if ((need_m(mode) && !U_obs_m) || (need_s(mode) && !U_obs_s)) return ET_ERR_INVALID_ARG;
    hipLaunchKernelGGL(scene_project_kernel, dim3(1), dim3(kTile), 0, (hipStream_t)stream, obs, (int)N, T_obs, k, U_obs_m,
                       U_obs_s, mode, static_dist, C_obs, nrm, obs_ori, flag);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_anchor_reconstruct_fwd(const float *C, int64_t N, int S, int k, int T_obs, int T_pred,
                                         const float *obs, const float *nrm, const float *A_m, const float *A_s,
                                         const float *U_pred_m, const float *U_pred_s, int mode, float static_dist,
                                         float *out, et_stream_t stream) {
    if (N < 0 || S < 1 || !dims_ok(T_obs, T_pred, k) || mode < 0 || mode > 3) return ET_ERR_INVALID_ARG;
    if (N == 0) return ET_OK;
    if (!C || !out || (!obs && !nrm && mode != ET_MODE_IDENTITY)) return ET_ERR_INVALID_ARG;
    if ((need_m(mode) && !U_pred_m) || (need_s(mode) && !U_pred_s)) return ET_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool fast = T_pred == 12 && k == 6 && S <= kTile && aligned16(out);
    if (fast) {
        const int TN = kTile / S;
        const size_t lds = sizeof(float) * ((size_t)TN * S * 24 + (size_t)TN * kNormStride + 2 * 24 * 6 + 2 * 6 * (size_t)S);
        const int64_t per_wg = (int64_t)TN * (S == 1 ? 1 : kReconTiles);
#ifdef ET_EXP_RECON_NOSTREAM  // measurement aid
        const bool stream = false;
#else
        const bool stream = S > 1 && N * S * (int64_t)96 > kStreamBytes;  // (S = 1: the default policy is faster, see kStreamBytes)
#endif
        auto kern = stream ? reconstruct_tile_kernel<12, 6, true> : reconstruct_tile_kernel<12, 6, false>;
        hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(N, per_wg)), dim3(kTile), lds, st, C, N, S, TN, T_obs, obs, nrm, A_m,
                           A_s, U_pred_m, U_pred_s, mode, static_dist, out);
    } else {
        const int64_t pairs = N * S;
        hipLaunchKernelGGL(reconstruct_generic_kernel, dim3((unsigned)ceil_div(pairs, kTile)), dim3(kTile), 0, st, C, N,
                           S, k, T_obs, T_pred, obs, nrm, A_m, A_s, U_pred_m, U_pred_s, mode, static_dist, out);
    }
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_anchor_reconstruct_metrics(const float *C, int64_t N, int S, int k, int T_obs, int T_pred,
                                             const float *obs, const float *nrm, const float *A_m, const float *A_s,
                                             const float *U_pred_m, const float *U_pred_s, int mode,
                                             float static_dist, const float *gt, float *ade, float *fde,
                                             et_stream_t stream) {
    return et_anchor_reconstruct_metrics_pose(C, N, S, k, T_obs, T_pred, obs, nrm, nullptr, A_m, A_s, U_pred_m, U_pred_s, mode,
                                              static_dist, gt, ade, fde, stream);
}

// the shapes reconstruct_metrics_mfma_kernel takes: a wavefront takes 64 / S trajectories per pass and normalises one
// ground-truth point per lane, so S <= 64 and (64 / S) * 12 <= 64, i.e. 12 <= S <= 64 (the model form is S = 20)
static bool metrics_mfma_shape(int64_t N, int S, int k, int T_pred, const float *gt) {
    return T_pred == 12 && k == 6 && aligned16(gt) && S >= 12 && S <= 64 && N >= 64 / S && N * S < ((int64_t)1 << 29) &&
           N * 24 < ((int64_t)1 << 30);
}

extern "C" int et_anchor_reconstruct_metrics_pose(const float *C, int64_t N, int S, int k, int T_obs, int T_pred,
                                                  const float *obs, const float *nrm, const float *pose, const float *A_m,
                                                  const float *A_s, const float *U_pred_m, const float *U_pred_s, int mode,
                                                  float static_dist, const float *gt, float *ade, float *fde,
                                                  et_stream_t stream) {
    if (N < 0 || S < 1 || !dims_ok(T_obs, T_pred, k) || mode < 0 || mode > 3) return ET_ERR_INVALID_ARG;
    if (N == 0) return ET_OK;
    // pose is consumed by the matrix-core kernel only: any other shape (or metrics_form = t) needs nrm or obs
    const bool pose_usable = pose && mode != ET_MODE_IDENTITY && metrics_mfma_shape(N, S, k, T_pred, gt) &&
                             options().metrics_form.load(std::memory_order_relaxed) != 't';
    if (!C || !gt || !ade || !fde || (!obs && !nrm && !pose_usable && mode != ET_MODE_IDENTITY)) return ET_ERR_INVALID_ARG;
    if ((need_m(mode) && !U_pred_m) || (need_s(mode) && !U_pred_s)) return ET_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool fast = T_pred == 12 && k == 6 && S <= kTile && aligned16(gt);
    if (fast) {
        const int TN = kTile / S;
        // option metrics_form (et_set_option; A/B runs, tests): t = the vector-ALU workgroup-tile kernel, f = fp32 matrix instructions only
        const int form = options().metrics_form.load(std::memory_order_relaxed);
        const int use_f16 = form != 'f';
        if (form != 't' && metrics_mfma_shape(N, S, k, T_pred, gt)) {
            const int TNW = 64 / S;
            const int64_t passes = ceil_div(N, TNW);
            const size_t lds = sizeof(float) * metrics_mfma_lds_floats(S, mode == ET_MODE_SPLIT ? 2 : 1);
            // a persistent grid: exactly the workgroups that are resident together (a surplus workgroup would start when
            // the others are done and double the run time of its CU)
#define ET_METRICS_LAUNCH(MODE, POSE)                                                                                     \
    do {                                                                                                                  \
        /* (asked once per mode and LDS size: the query is a few microseconds of host time, as much as a scene call's   \
           whole kernel) */                                                                                               \
        static std::atomic<unsigned long long> cached{0};                                                                 \
        int per_cu = 0;                                                                                                   \
        const unsigned long long c = cached.load(std::memory_order_relaxed);                                             \
        if ((c >> 32) == (unsigned long long)lds + 1) per_cu = (int)(c & 0xffffffffull);                                  \
        else {                                                                                                            \
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reconstruct_metrics_mfma_kernel<12, 6, MODE, POSE>, \
                                                             kMetWaves * 64, lds) != hipSuccess || per_cu < 1)           \
                per_cu = 2;                                                                                               \
            cached.store((((unsigned long long)lds + 1) << 32) | (unsigned)per_cu, std::memory_order_relaxed);            \
        }                                                                                                                 \
        const unsigned g = (unsigned)min((int64_t)cu_count() * per_cu, ceil_div(passes, kMetWaves));                      \
        hipLaunchKernelGGL((reconstruct_metrics_mfma_kernel<12, 6, MODE, POSE>), dim3(g), dim3(kMetWaves * 64), lds, st,  \
                           C, (int)N, S, TNW, T_obs, obs, POSE ? pose : nrm, A_m, A_s, U_pred_m, U_pred_s, static_dist,   \
                           gt, ade, fde, use_f16);                                                                        \
    } while (0)
            if (pose_usable) {
                if (mode == ET_MODE_SPLIT) ET_METRICS_LAUNCH(ET_MODE_SPLIT, true);
                else if (mode == ET_MODE_MOVING) ET_METRICS_LAUNCH(ET_MODE_MOVING, true);
                else ET_METRICS_LAUNCH(ET_MODE_STATIC, true);
            } else if (mode == ET_MODE_SPLIT) ET_METRICS_LAUNCH(ET_MODE_SPLIT, false);
            else if (mode == ET_MODE_MOVING) ET_METRICS_LAUNCH(ET_MODE_MOVING, false);
            else if (mode == ET_MODE_STATIC) ET_METRICS_LAUNCH(ET_MODE_STATIC, false);
            else ET_METRICS_LAUNCH(ET_MODE_IDENTITY, false);
#undef ET_METRICS_LAUNCH
        } else {
            const size_t lds = sizeof(float) * ((size_t)TN * 24 + (size_t)TN * S * 2 + (size_t)TN * kNormStride + 2 * 24 * 6 +
                                                2 * 6 * (size_t)S);
            hipLaunchKernelGGL((reconstruct_metrics_tile_kernel<12, 6>), dim3((unsigned)ceil_div(N, TN)), dim3(kTile), lds, st,
                               C, N, S, TN, T_obs, obs, nrm, A_m, A_s, U_pred_m, U_pred_s, mode, static_dist, gt, ade, fde);
        }
    } else {
        hipLaunchKernelGGL(reconstruct_metrics_generic_kernel, dim3((unsigned)ceil_div(N, kTile)), dim3(kTile), 0, st, C, N,
                           S, k, T_obs, T_pred, obs, nrm, A_m, A_s, U_pred_m, U_pred_s, mode, static_dist, gt, ade, fde);
    }
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_anchor_reconstruct_bwd(const float *dtraj, int64_t N, int S, int k, int T_obs, int T_pred,
                                         const float *obs, const float *nrm, const float *U_pred_m,
                                         const float *U_pred_s, int mode, float static_dist, float *dC,
                                         et_stream_t stream) {
    if (N < 0 || S < 1 || !dims_ok(T_obs, T_pred, k) || mode < 0 || mode > 3) return ET_ERR_INVALID_ARG;
    if (N == 0) return ET_OK;
    if (!dtraj || !dC || (!obs && !nrm && mode != ET_MODE_IDENTITY)) return ET_ERR_INVALID_ARG;
    if ((need_m(mode) && !U_pred_m) || (need_s(mode) && !U_pred_s)) return ET_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const bool fast = T_pred == 12 && k == 6 && S <= kTile && aligned16(dtraj);
    if (fast) {
        const int TN = kTile / S;
        const size_t lds = sizeof(float) * ((size_t)TN * S * 24 + (size_t)TN * kNormStride + 2 * 24 * 6 + 4);  // + a spare float4
        hipLaunchKernelGGL((reconstruct_bwd_tile_kernel<12, 6>), dim3((unsigned)ceil_div(N, TN)), dim3(kTile), lds, st,
                           dtraj, N, S, TN, T_obs, obs, nrm, U_pred_m, U_pred_s, mode, static_dist, dC);
    } else {
        const int64_t pairs = N * S;
        hipLaunchKernelGGL(reconstruct_bwd_generic_kernel, dim3((unsigned)ceil_div(pairs, kTile)), dim3(kTile), 0, st,
                           dtraj, N, S, k, T_obs, T_pred, obs, nrm, U_pred_m, U_pred_s, mode, static_dist, dC);
    }
    ET_LAUNCH_CHECK();
    return ET_OK;
}

#ifdef ET_EXP_METSTAMP
extern "C" int et_debug_metstamp(unsigned long long *host, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(et::g_metstamp), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(et::g_metstamp), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
#endif
