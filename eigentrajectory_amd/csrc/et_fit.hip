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
constexpr int kFitTile = 128;  // trajectories per workgroup pass

// ------------------------------------------------------------------------------------------
// Gram kernel, specialised on (T_obs, T_pred).
//  1. coalesced float4 loads of kFitTile rows -> LDS (raw fp32)
//  2. lane = trajectory: normalise, write the 2T_obs + 2T_pred features as fp64 to LDS
//     (rows that do not belong to descriptor `which` are written as zeros)
//  3. 8 groups of 32 lanes; lane g of a group owns one 4x4 block of the upper triangle of
//     G_obs (10 blocks) or G_pred (21 blocks) and walks the group's trajectories:
//     2 x 32-byte LDS reads -> 16 fp64 FMAs in registers.
//  4. groups are summed through LDS; every workgroup writes one partial (fixed slot).
// A grid-stride loop over tiles keeps the number of partials small.
// ------------------------------------------------------------------------------------------
template <int TO, int TP>
struct GramLayout {
    static constexpr int DO = 2 * TO, DP = 2 * TP, D = DO + DP;
    static constexpr int BO = DO / 4, BP = DP / 4;                   // 4-wide blocks per side
    static constexpr int NBO = BO * (BO + 1) / 2, NBP = BP * (BP + 1) / 2;
    static constexpr int NB = NBO + NBP;                             // upper-triangular 4x4 blocks
    static constexpr int kPartial = NB * 16 + 1;                     // doubles per partial (+ row count)
};

template <int TO, int TP>
__global__ __launch_bounds__(kFitThreads) void gram_tile_kernel(
    const float *__restrict__ obs, const float *__restrict__ pred, int64_t N, int mode, float static_dist, int which,
    double *__restrict__ partials) {
    using L = GramLayout<TO, TP>;
    constexpr int DO = L::DO, DP = L::DP, D = L::D, NB = L::NB;
    constexpr int QO = DO / 4, QP = DP / 4;
    constexpr int PO = QO + 1, PP = QP + 1;
    constexpr int FP = D + 2;  // fp64 feature row pitch (doubles); +2 keeps 16-B alignment, spreads banks
    static_assert(NB <= 32, "one 4x4 block per lane of a 32-lane group");

    // one LDS block: [fp64 features | raw obs rows | raw pred rows]; the group partials of the
    // final reduction reuse the feature area.
    constexpr int kFeatDoubles = kFitTile * FP;
    constexpr int kRedDoubles = (kFitThreads / 32) * NB * 16;
    constexpr int kHeadDoubles = kFeatDoubles > kRedDoubles ? kFeatDoubles : kRedDoubles;
    __shared__ __attribute__((aligned(16))) double sMem[kHeadDoubles + 2 * kFitTile * (PO + PP) + 2];
    double *sFeat = sMem;
    double *sRed = sMem;
    float4 *sObs = reinterpret_cast<float4 *>(sMem + kHeadDoubles);
    float4 *sPred = sObs + kFitTile * PO;
    int *sCountPtr = reinterpret_cast<int *>(sPred + kFitTile * PP);

    const int tid = threadIdx.x;
    const int grp = tid >> 5, g = tid & 31;

    // block (bi,bj) owned by lane g: first the NBO blocks of G_obs, then the NBP of G_pred
    int off_i = 0, off_j = 0;
    bool active = g < NB;
    {
        int b = g, base = 0, nb = L::BO;
        if (g >= L::NBO) {
            b = g - L::NBO;
            base = DO;
            nb = L::BP;
        }
        int bi = 0;
        while (active && b >= nb - bi) {
            b -= nb - bi;
            ++bi;
        }
        off_i = base + 4 * bi;
        off_j = base + 4 * (bi + b);
    }

    double acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0;
    int my_count = 0;

    const int64_t n_tiles = ceil_div(N, (int64_t)kFitTile);
    // software pipeline: the raw rows of the NEXT tile are fetched into registers while the current
    // tile is normalised and accumulated, so the HBM latency is paid once, not once per tile
    constexpr int NLO = (kFitTile * QO + kFitThreads - 1) / kFitThreads;  // float4 of obs per thread
    constexpr int NLP = (kFitTile * QP + kFitThreads - 1) / kFitThreads;  // float4 of pred per thread
    float4 ro[NLO], rp[NLP];
    auto fetch = [&](int64_t tile) {
        const int64_t n0 = tile * kFitTile;
        const int rows = (int)min((int64_t)kFitTile, N - n0);
        const float4 *go = reinterpret_cast<const float4 *>(obs + n0 * DO);
        const float4 *gp = reinterpret_cast<const float4 *>(pred + n0 * DP);
#pragma unroll
        for (int j = 0; j < NLO; ++j) {
            const int q = tid + j * kFitThreads;
            ro[j] = q < rows * QO ? go[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < NLP; ++j) {
            const int q = tid + j * kFitThreads;
            rp[j] = q < rows * QP ? gp[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if ((int64_t)blockIdx.x < n_tiles) fetch(blockIdx.x);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t n0 = tile * kFitTile;
        const int rows = (int)min((int64_t)kFitTile, N - n0);
        __syncthreads();  // previous pass done with sFeat / sObs
#pragma unroll
        for (int j = 0; j < NLO; ++j) {
            const int q = tid + j * kFitThreads;
            if (q < kFitTile * QO) sObs[(q / QO) * PO + (q % QO)] = ro[j];
        }
#pragma unroll
        for (int j = 0; j < NLP; ++j) {
            const int q = tid + j * kFitThreads;
            if (q < kFitTile * QP) sPred[(q / QP) * PP + (q % QP)] = rp[j];
        }
        if (tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);
        __syncthreads();
        if (tid < kFitTile) {
            double *f = sFeat + tid * FP;
            bool use = false;
            if (tid < rows) {
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
                const RowNorm p = row_norm(ox, oy, ox - xo[DO - 6], oy - xo[DO - 5], mode, static_dist);
                use = p.mv == which;
                if (use) {
#pragma unroll
                    for (int t = 0; t < TO; ++t) {
                        float a, b;
                        normalize_point(p, xo[2 * t], xo[2 * t + 1], a, b);
                        f[2 * t] = (double)a;
                        f[2 * t + 1] = (double)b;
                    }
#pragma unroll
                    for (int q = 0; q < QP; ++q) {
                        const float4 v = sPred[tid * PP + q];
                        float a, b;
                        normalize_point(p, v.x, v.y, a, b);
                        f[DO + 4 * q] = (double)a;
                        f[DO + 4 * q + 1] = (double)b;
                        normalize_point(p, v.z, v.w, a, b);
                        f[DO + 4 * q + 2] = (double)a;
                        f[DO + 4 * q + 3] = (double)b;
                    }
                    ++my_count;
                }
            }
            if (!use) {
#pragma unroll
                for (int e = 0; e < D; ++e) f[e] = 0.0;
            }
        }
        __syncthreads();
        if (active) {
            // group grp walks trajectories grp, grp+8, ... of the tile
            for (int r = grp; r < kFitTile; r += kFitThreads / 32) {
                const double *f = sFeat + r * FP;
                double xi[4], xj[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xi[e] = f[off_i + e];
                    xj[e] = f[off_j + e];
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a * 4 + b] = fma(xi[a], xj[b], acc[a * 4 + b]);
            }
        }
    }

    // ---- combine the 8 groups (fixed order), one partial per workgroup
    __syncthreads();  // all groups finished reading sFeat (aliased by sRed)
    if (tid == 0) *sCountPtr = 0;
    __syncthreads();
    if (active) {
#pragma unroll
        for (int e = 0; e < 16; ++e) sRed[(grp * NB + g) * 16 + e] = acc[e];
    }
    if (my_count) atomicAdd(sCountPtr, my_count);
    __syncthreads();
    double *dst = partials + (size_t)blockIdx.x * L::kPartial;
    for (int i = tid; i < NB * 16; i += kFitThreads) {
        double s = 0.0;
        for (int gr = 0; gr < kFitThreads / 32; ++gr) s += sRed[gr * NB * 16 + i];
        dst[i] = s;
    }
    if (tid == 0) dst[NB * 16] = (double)*sCountPtr;
}

// Sum the workgroup partials of one entry: one wavefront per entry, a fixed strided + butterfly
// order (reproducible for a given grid).
__global__ __launch_bounds__(64) void gram_reduce_kernel(const double *__restrict__ partials, int n_partials, int stride,
                                                         double *__restrict__ sums) {
    const int e = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < n_partials; b += 64) s += partials[(size_t)b * stride + e];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) sums[e] = s;
}

// Expand the summed 4x4 upper-triangular blocks into the two full symmetric matrices.  One workgroup.
template <int TO, int TP>
__global__ __launch_bounds__(kFitThreads) void gram_finish_kernel(const double *__restrict__ sSum,
                                                                  double *__restrict__ G_obs,
                                                                  double *__restrict__ G_pred,
                                                                  int64_t *__restrict__ count) {
    using L = GramLayout<TO, TP>;
    constexpr int DO = L::DO, DP = L::DP, NB = L::NB;
    for (int b = threadIdx.x; b < NB; b += kFitThreads) {
        int bb = b, nb = L::BO, dim = DO;
        double *G = G_obs;
        if (b >= L::NBO) {
            bb = b - L::NBO;
            nb = L::BP;
            dim = DP;
            G = G_pred;
        }
        int bi = 0;
        while (bb >= nb - bi) {
            bb -= nb - bi;
            ++bi;
        }
        const int bj = bi + bb;
        for (int a = 0; a < 4; ++a)
            for (int c = 0; c < 4; ++c) {
                const double v = sSum[b * 16 + a * 4 + c];
                const int i = 4 * bi + a, j = 4 * bj + c;
                if (bi == bj) {
                    if (j >= i) {  // mirror the upper triangle of a diagonal block: exactly symmetric output
                        G[i * dim + j] = v;
                        G[j * dim + i] = v;
                    }
                } else {
                    G[i * dim + j] = v;
                    G[j * dim + i] = v;
                }
            }
    }
    if (threadIdx.x == 0) *count = (int64_t)sSum[NB * 16];
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
// A round applies n/2 disjoint rotations: lanes compute the (c, s) of one pair each, then all
// lanes sweep the row updates of every pair, then the column updates (A and V): three barriers
// per round instead of four per rotation.  Same pairing, same formulas and the same per-element
// operation order as the CPU oracle (oracle/et_oracle.c: eto_jacobi) => bit-identical output.
// 24 x 24 converges in ~8 sweeps (23 rounds each).
// ------------------------------------------------------------------------------------------
constexpr int kJacobiMaxSweeps = 30;
constexpr int kEighThreads = 256;  // 4 wavefronts share the element updates of a round

__global__ __launch_bounds__(kEighThreads) void eigh_topk_kernel(const double *__restrict__ G, int n, int k,
                                                       float *__restrict__ U, float *__restrict__ sigma) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *A = sm;                  // n*n
    double *V = sm + n * n;          // n*n
    double *sC = V + n * n;          // 32 cosines
    double *sS = sC + 32;            // 32 sines
    int *sP = reinterpret_cast<int *>(sS + 32 + 2 * (kEighThreads / 64));  // 32 p, 32 q, 32 active, 64 used, 1 flag
    int *sQ = sP + 32, *sAct = sQ + 32, *sUsed = sAct + 32;
    int &sFlag = sUsed[64];
    const int lane = threadIdx.x;
    const int m = (n + 1) & ~1, half = m / 2;
    for (int i = lane; i < n * n; i += kEighThreads) {
        A[i] = G[i];
        V[i] = (i / n == i % n) ? 1.0 : 0.0;
    }
    __syncthreads();
    // element slots of this thread in the row / column phases: e = lane + t*256 -> (pair i, index j)
    constexpr int kSlots = (32 * 64 + kEighThreads - 1) / kEighThreads;
    int slot_i[kSlots], slot_j[kSlots];
#pragma unroll
    for (int t = 0; t < kSlots; ++t) {
        const int e = lane + t * kEighThreads;
        slot_i[t] = e < half * n ? e / n : -1;
        slot_j[t] = e < half * n ? e % n : 0;
    }
    double *sMax = sS + 32;  // 2 * (kEighThreads / 64) partial maxima
    for (int sweep = 0; sweep < kJacobiMaxSweeps; ++sweep) {
        // converged when max |off-diagonal| <= 1e-15 max |diagonal| (maxima: order independent)
        double off = 0.0, diag = 0.0;
        for (int e = lane; e < n * n; e += kEighThreads) {
            const int i = e / n, j = e - i * n;
            const double a = fabs(A[e]);
            if (i == j) diag = a > diag ? a : diag;
            else if (j > i) off = a > off ? a : off;
        }
        for (int o = 32; o > 0; o >>= 1) {
            const double po = __shfl_xor(off, o), pd = __shfl_xor(diag, o);
            off = po > off ? po : off;
            diag = pd > diag ? pd : diag;
        }
        if ((lane & 63) == 0) {
            sMax[2 * (lane >> 6)] = off;
            sMax[2 * (lane >> 6) + 1] = diag;
        }
        __syncthreads();
        if (lane == 0) {
            for (int w = 1; w < kEighThreads / 64; ++w) {
                off = sMax[2 * w] > off ? sMax[2 * w] : off;
                diag = sMax[2 * w + 1] > diag ? sMax[2 * w + 1] : diag;
            }
            sFlag = (off <= 1e-15 * diag) ? 1 : 0;
        }
        __syncthreads();
        if (sFlag) break;
        for (int r = 0; r < m - 1; ++r) {
            if (lane < half) {
                int a, b;
                if (lane == 0) {
                    a = m - 1;
                    b = r;
                } else {
                    a = (r + lane) % (m - 1);
                    b = (r + (m - 1) - lane) % (m - 1);
                }
                const int p = a < b ? a : b, q = a < b ? b : a;
                int act = 0;
                if (q < n) {
                    const double apq = A[p * n + q];
                    if (apq != 0.0) {
                        const double app = A[p * n + p], aqq = A[q * n + q];
                        const double theta = (aqq - app) / (2.0 * apq);
                        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        const double c = 1.0 / sqrt(t * t + 1.0);
                        sC[lane] = c;
                        sS[lane] = t * c;
                        act = 1;
                    }
                }
                sP[lane] = p;
                sQ[lane] = q;
                sAct[lane] = act;
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < kSlots; ++t) {  // rows p,q of every pair
                const int i = slot_i[t], j = slot_j[t];
                if (i >= 0 && sAct[i]) {
                    const int p = sP[i], q = sQ[i];
                    const double c = sC[i], s = sS[i];
                    const double apj = A[p * n + j], aqj = A[q * n + j];
                    A[p * n + j] = c * apj - s * aqj;
                    A[q * n + j] = s * apj + c * aqj;
                }
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < kSlots; ++t) {  // columns p,q of every pair, A and V
                const int i = slot_i[t], j = slot_j[t];
                if (i >= 0 && sAct[i]) {
                    const int p = sP[i], q = sQ[i];
                    const double c = sC[i], s = sS[i];
                    const double ajp = A[j * n + p], ajq = A[j * n + q];
                    A[j * n + p] = c * ajp - s * ajq;
                    A[j * n + q] = s * ajp + c * ajq;
                    const double vjp = V[j * n + p], vjq = V[j * n + q];
                    V[j * n + p] = c * vjp - s * vjq;
                    V[j * n + q] = s * vjp + c * vjq;
                }
            }
            __syncthreads();
            if (lane < half && sAct[lane]) {  // (the next round's pairs are different entries)
                A[sP[lane] * n + sQ[lane]] = 0.0;
                A[sQ[lane] * n + sP[lane]] = 0.0;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (lane < 64) sUsed[lane] = 0;
    __syncthreads();
    for (int j = 0; j < k; ++j) {
        if (lane == 0) {
            int best = -1;
            for (int i = 0; i < n; ++i)
                if (!sUsed[i] && (best < 0 || A[i * n + i] > A[best * n + best])) best = i;
            sUsed[best] = 1;
            int im = 0;
            for (int i = 1; i < n; ++i)
                if (fabs(V[i * n + best]) > fabs(V[im * n + best])) im = i;
            sFlag = best | ((V[im * n + best] < 0.0 ? 1 : 0) << 8);
            const double lam = A[best * n + best];
            sigma[j] = (float)sqrt(lam > 0.0 ? lam : 0.0);
        }
        __syncthreads();
        const int best = sFlag & 0xff;
        const double sgn = (sFlag >> 8) ? -1.0 : 1.0;
        if (lane < n) U[lane * k + j] = (float)(sgn * V[lane * n + best]);
        __syncthreads();
    }
}

static int fit_grid(int64_t N) {
    // enough workgroups to fill 256 CUs twice, few enough that the partial reduction stays trivial
    const int64_t tiles = ceil_div(N, (int64_t)kFitTile);
    return (int)(tiles < 512 ? (tiles > 0 ? tiles : 1) : 512);
}

}  // namespace et

using namespace et;

extern "C" size_t et_fit_gram_workspace_bytes(int64_t N, int T_obs, int T_pred) {
    const size_t DO = 2 * (size_t)T_obs, DP = 2 * (size_t)T_pred;
    const size_t per = DO * DO + DP * DP + 1;  // the generic layout is the larger one
    return sizeof(double) * per * ((size_t)fit_grid(N) + 1);  // workgroup partials + their sums
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
        hipLaunchKernelGGL((gram_tile_kernel<8, 12>), dim3(grid), dim3(kFitThreads), 0, st, obs, pred, N, mode,
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

extern "C" int et_eigh_topk(const double *G, int n, int k, float *U, float *sigma, et_stream_t stream) {
    if (!G || !U || !sigma || n < 1 || n > 64 || k < 1 || k > n) return ET_ERR_INVALID_ARG;
    const size_t lds = sizeof(double) * (2 * (size_t)n * n + 64 + 2 * (kEighThreads / 64)) + sizeof(int) * (32 * 3 + 64 + 2);
    if (lds > 48 * 1024)
        ET_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(eigh_topk_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(eigh_topk_kernel, dim3(1), dim3(kEighThreads), lds, (hipStream_t)stream, G, n, k, U, sigma);
    ET_LAUNCH_CHECK();
    return ET_OK;
}
