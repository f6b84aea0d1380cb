// et_kmeanspp.hip -- the seeding and pre-processing half of the reference's anchor clustering,
//
//     sklearn.cluster.KMeans(n_clusters=S, random_state=0, init='k-means++', n_init=10)
//                                                          (EigenTrajectory/anchor.py:65-71),
//
// on the device.  sklearn is a third-party dependency of the reference (version unpinned; the
// golden fixtures were captured with 1.7.2); what is restated here is its published algorithm,
// in the arithmetic its float32 code path uses, so that the device recipe draws the SAME seed
// points from the same RandomState stream:
//
//   * KMeans.fit mean-centres the data with `X -= X.mean(axis=0)` and derives
//     `tol = 1e-4 * mean(var(X, axis=0))`: numpy reduces axis 0 of a C-ordered (N,d) float32
//     array by adding the rows ONE AFTER THE OTHER in float32 (no pairwise tree on that axis).
//     `colstats_kernel` reproduces that order: one lane per coordinate, rows staged through LDS.
//   * greedy k-means++ (Arthur & Vassilvitskii 2007, with 2 + log K candidates per centre):
//     squared distances are evaluated in float64 as ((-2 x.c) + |c|^2) + |x|^2, cast to
//     float32 and clamped at 0 (sklearn's `_euclidean_distances_upcast` for float32 input); a
//     candidate is the first index whose float64 running sum of the current closest distances
//     reaches `u * potential` (`np.searchsorted(stable_cumsum(closest), u * pot)`), and the
//     candidate with the smallest new potential wins.
//     The potential itself is a float32 BLAS dot in sklearn (unspecified summation order); here
//     it is the float64 sum rounded to float32, i.e. the value sklearn's dot approximates.
//
// Everything is enqueued on the caller's stream without host synchronisation: the uniforms of the
// whole seeding are drawn on the host beforehand (the stream consumption is fixed: one draw for
// the first centre, n_trials per further centre) and read from device memory.
//
// Kernels
//   colstats_kernel        sequential fp32 column sums (of x, or of (x - a)^2)
//   center_kernel          X[r][n] -= mean[r]
//   kpp_blocksum_kernel    closest <- D[best] (copy) + fp64 block sums of closest
//   kpp_locate_kernel      block prefix + in-block search of every threshold -> candidate indices
//   kpp_dist_kernel        float64 distances to the candidates, min with closest, potentials
//   kpp_select_kernel      arg-min potential -> new centre, its index, the new potential
#include <cstdlib>
#include <type_traits>

#include "et_common.h"

namespace et {

constexpr int kPpThreads = 256;
constexpr int kPpBlock = 4096;    // elements per block of the running-sum search
constexpr int kPpMaxTrials = 8;   // 2 + log(255) = 7
__host__ __device__ static inline int colstats_chunk(int d) {  // rows staged per LDS tile in colstats_kernel
    const int c = (6144 / d) & ~3;
    return c > 1024 ? 1024 : c;
}
static inline size_t colstats_lds_bytes(int d) { return sizeof(float) * 2 * (size_t)d * (colstats_chunk(d) + 4); }

// ------------------------------------------------------------------------------------------
// numpy's add.reduce(axis=0) on a C-ordered (N,d) float32 array: out[j] += x[i][j], i ascending.
// OP 0: sum of x;  OP 1: sum of (x - shift[j])^2   (np.var's second pass).  X is d-major (d,N).
// out[j] = sum / N (float32 division, like umr_sum / true_divide in np.mean / np.var).
// The sum is inherently serial (every fp32 addition rounds), so what can be fast is everything around the one dependent
// v_add_f32 per row: wavefront 0 (one lane per coordinate) adds chunk i from LDS, four rows per ds_read_b128, while
// wavefronts 1..3 stage chunk i + 1 into the other half of the LDS tile -- one barrier per 2048 rows.  (The first
// version staged 512 rows per barrier pair with all threads and added them with scalar LDS reads: 0.8 ms per pass at
// N = 6e4, a third of an anchor clustering once its ten initialisations ran side by side.)
template <int OP>
__global__ __launch_bounds__(kPpThreads) void colstats_kernel(const float *__restrict__ X, int64_t N, int d,
                                                               const float *__restrict__ shift,
                                                               float *__restrict__ out) {
    // rows per tile: 1024 for d <= 6, fewer for wider points (two tiles of d x (rows + 4) floats in ~48 KB of LDS)
    const int kColChunk = colstats_chunk(d), kPitch = kColChunk + 4;  // (16-byte aligned rows, a different bank per coordinate)
    extern __shared__ __attribute__((aligned(16))) float colstats_lds[];
    float *tile[2] = {colstats_lds, colstats_lds + d * kPitch};
    const int tid = threadIdx.x;
    float acc = 0.0f;
    const float a = (OP == 1 && tid < d) ? shift[tid] : 0.0f;
    const int64_t n_chunks = (N + kColChunk - 1) / kColChunk;
    auto stage = [&](int64_t chunk, int first, int step) {  // rows of `chunk` -> tile[chunk & 1], by threads first, first + step, ...
        const int64_t base = chunk * kColChunk;
        const int cnt = (int)((N - base) < kColChunk ? (N - base) : kColChunk);
        float *dst = tile[chunk & 1];
        for (int idx = first; idx < d * kColChunk; idx += step) {
            const int r = idx / kColChunk, c = idx - r * kColChunk;
            dst[r * kPitch + c] = c < cnt ? X[(int64_t)r * N + base + c] : 0.0f;
        }
    };
    stage(0, tid, kPpThreads);
    __syncthreads();
    for (int64_t chunk = 0; chunk < n_chunks; ++chunk) {
        if (tid >= kWave) {
            if (chunk + 1 < n_chunks) stage(chunk + 1, tid - kWave, kPpThreads - kWave);
        } else if (tid < d) {
            const int64_t base = chunk * kColChunk;
            const int cnt = (int)((N - base) < kColChunk ? (N - base) : kColChunk);
            const float4 *row = reinterpret_cast<const float4 *>(tile[chunk & 1] + tid * kPitch);
            const int full = cnt / 4;
#pragma unroll 8
            for (int q = 0; q < full; ++q) {
                float4 v = row[q];
                if (OP == 1) {
                    const float t0 = v.x - a, t1 = v.y - a, t2 = v.z - a, t3 = v.w - a;
                    v = make_float4(t0 * t0, t1 * t1, t2 * t2, t3 * t3);
                }
                acc = acc + v.x;
                acc = acc + v.y;
                acc = acc + v.z;
                acc = acc + v.w;
            }
            const float *tail = tile[chunk & 1] + tid * kPitch;
            for (int c = 4 * full; c < cnt; ++c) {
                float v = tail[c];
                if (OP == 1) {
                    const float t = v - a;
                    v = t * t;
                }
                acc = acc + v;
            }
        }
        __syncthreads();
    }
    if (tid < d) out[tid] = acc / (float)N;
}

__global__ __launch_bounds__(kPpThreads) void center_kernel(float *__restrict__ X, int64_t N, int d,
                                                             const float *__restrict__ mean) {
    const int64_t total = N * d;
    for (int64_t i = (int64_t)blockIdx.x * kPpThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kPpThreads)
        X[i] = X[i] - mean[i / N];
}

// tol = mean(var) * 1e-4 in float32 (sklearn _tolerance: np.mean(variances) * tol, a float32 scalar times a
// Python float -> float32 arithmetic)
__global__ void tolerance_kernel(const float *__restrict__ var, int d, float rel_tol, float *__restrict__ tol_out) {
    if (threadIdx.x == 0) {
        float s = 0.0f;
        for (int j = 0; j < d; ++j) s = s + var[j];
        tol_out[0] = (s / (float)d) * rel_tol;
    }
}

// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// blockIdx.y of the four kernels below = one of several seedings of the SAME points run side by side (the n_init
// initialisations of the sklearn recipe): every scratch pointer moves on by y * ws bytes, the draws / outputs by their
// element strides.  {0, 0, 0, 0} for a single seeding.
struct PpBatch {
    int64_t ws, uniforms, centers, indices;
};
template <typename T>
__device__ __forceinline__ T *pp_shift(T *p, int64_t bytes) {
    return p ? reinterpret_cast<T *>(reinterpret_cast<char *>(const_cast<typename std::remove_const<T>::type *>(p)) + bytes) : p;
}

// closest[n] <- D[best][n] (when D != nullptr), blocksums[b] = float64 sum of the block's closest values.
__global__ __launch_bounds__(kPpThreads) void kpp_blocksum_kernel(const float *__restrict__ D, const int *__restrict__ best,
                                                                   float *__restrict__ closest, int64_t N,
                                                                   double *__restrict__ blocksums, PpBatch bt) {
    {
        const int64_t off = (int64_t)blockIdx.y * bt.ws;
        D = pp_shift(D, off), best = pp_shift(best, off), closest = pp_shift(closest, off), blocksums = pp_shift(blocksums, off);
    }
    __shared__ double part[kPpThreads / kWave];
    const float *src = D ? D + (int64_t)best[0] * N : closest;
    const int64_t base = (int64_t)blockIdx.x * kPpBlock;
    double s = 0.0;
    for (int i = threadIdx.x; i < kPpBlock; i += kPpThreads) {
        const int64_t n = base + i;
        if (n < N) {
            const float v = src[n];
            if (D) closest[n] = v;
            s += (double)v;
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x / kWave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kPpThreads / kWave; ++w) t += part[w];
        blocksums[blockIdx.x] = t;
    }
}

// One workgroup.  c == 0: cand[0] = the first centre's index, floor(u0 * N) (RandomState.choice with uniform p).
// c >= 1: thresholds t_j = u_j * (double)pot; cand[j] = first n with cumsum(closest)[n] >= t_j, clipped to N-1.
__global__ __launch_bounds__(kPpThreads) void kpp_locate_kernel(const float *__restrict__ closest, int64_t N,
                                                                 const double *__restrict__ blocksums, int64_t nb,
                                                                 double *__restrict__ prefix,
                                                                 const double *__restrict__ uniforms, int n_trials, int c,
                                                                 const float *__restrict__ pot, int64_t *__restrict__ cand,
                                                                 PpBatch bt) {
    {
        const int64_t off = (int64_t)blockIdx.y * bt.ws;
        closest = pp_shift(closest, off), blocksums = pp_shift(blocksums, off), prefix = pp_shift(prefix, off);
        pot = pp_shift(pot, off), cand = pp_shift(cand, off);
        uniforms += (int64_t)blockIdx.y * bt.uniforms;
    }
    if (c == 0) {
        if (threadIdx.x == 0) {
            int64_t f = (int64_t)(uniforms[0] * (double)N);
            cand[0] = f < N - 1 ? f : N - 1;
        }
        return;
    }
    if (threadIdx.x == 0) {  // inclusive running sum over the blocks, left to right like np.cumsum
        double run = 0.0;
        for (int64_t b = 0; b < nb; ++b) {
            run += blocksums[b];
            prefix[b] = run;
        }
    }
    __syncthreads();
    const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    for (int j = wave; j < n_trials; j += kPpThreads / kWave) {
        const double t = uniforms[j] * (double)pot[0];
        // first block whose inclusive prefix reaches t
        int64_t lo = 0, hi = nb;  // answer in [lo, hi]; hi == nb: beyond the end
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (prefix[mid] >= t) hi = mid;
            else lo = mid + 1;
        }
        if (lo >= nb) {
            if (lane == 0) cand[j] = N - 1;  // np.clip(candidate_ids, None, n - 1)
            continue;
        }
        const int64_t base = lo * kPpBlock;
        const double before = lo > 0 ? prefix[lo - 1] : 0.0;
        constexpr int per = kPpBlock / kWave;  // 64 consecutive elements per lane
        double mine = 0.0;
        for (int i = 0; i < per; ++i) {
            const int64_t n = base + (int64_t)lane * per + i;
            if (n < N) mine += (double)closest[n];
        }
        double incl = mine;  // inclusive scan over the lanes
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const double up = __shfl_up(incl, off, kWave);
            if (lane >= off) incl += up;
        }
        const double excl = before + (incl - mine);
        const bool crosses = (before + incl) >= t;
        const unsigned long long ballot = __ballot(crosses);
        int64_t found;
        if (ballot == 0ull) {  // rounding: the block total reached t but the lane sums fall a hair short
            const int64_t last = base + kPpBlock - 1;
            found = last < N - 1 ? last : N - 1;
        } else {
            const int owner = __ffsll((long long)ballot) - 1;
            found = -1;
            if (lane == owner) {
                double run = excl;
                int64_t n = base + (int64_t)lane * per;
                found = n + per - 1 < N - 1 ? n + per - 1 : N - 1;
                for (int i = 0; i < per && n + i < N; ++i) {
                    run += (double)closest[n + i];
                    if (run >= t) {
                        found = n + i;
                        break;
                    }
                }
            }
            found = __shfl(found, owner, kWave);
        }
        if (lane == 0) cand[j] = found;
    }
}

// D[j][n] = min(closest[n], max((float)(((-2 x_n.c_j) + |c_j|^2) + |x_n|^2), 0)); partials[j][block] = sum_n D[j][n] (fp64).
// closest == nullptr (first centre): no min.
// DIM = 6: the coefficient dimension of the path, compile-time loops and the point in registers; DIM = 0: any
// d <= ET_KMEANS_MAX_D with runtime loops (its runtime-indexed `double x[d]` lives in scratch memory)
template <int DIM>
__global__ __launch_bounds__(kPpThreads) void kpp_dist_kernel(const float *__restrict__ X, int64_t N, int d_rt,
                                                               const int64_t *__restrict__ cand, int n_trials,
                                                               const float *__restrict__ closest, float *__restrict__ D,
                                                               double *__restrict__ partials, int64_t n_blocks, PpBatch bt) {
    {
        const int64_t off = (int64_t)blockIdx.y * bt.ws;
        cand = pp_shift(cand, off), closest = pp_shift(closest, off), D = pp_shift(D, off), partials = pp_shift(partials, off);
    }
    constexpr int DM = DIM ? DIM : ET_KMEANS_MAX_D;
    const int d = DIM ? DIM : d_rt;
    __shared__ double cen[kPpMaxTrials][ET_KMEANS_MAX_D + 1];  // [j][r], [j][d] = |c_j|^2
    __shared__ double part[kPpMaxTrials][kPpThreads / kWave];
    if (threadIdx.x < n_trials * d) {
        const int j = threadIdx.x / d, r = threadIdx.x - j * d;
        cen[j][r] = (double)X[(int64_t)r * N + cand[j]];
    }
    __syncthreads();
    if (threadIdx.x < n_trials) {
        double cc = 0.0;
        for (int r = 0; r < d; ++r) cc += cen[threadIdx.x][r] * cen[threadIdx.x][r];
        cen[threadIdx.x][d] = cc;
    }
    __syncthreads();
    double sums[kPpMaxTrials];
#pragma unroll
    for (int j = 0; j < kPpMaxTrials; ++j) sums[j] = 0.0;
    const int64_t base = (int64_t)blockIdx.x * kPpBlock;
    for (int i = threadIdx.x; i < kPpBlock; i += kPpThreads) {
        const int64_t n = base + i;
        if (n >= N) break;
        double x[DM];
        double xx = 0.0;
        if constexpr (DIM != 0) {
#pragma unroll
            for (int r = 0; r < DM; ++r) {
                x[r] = (double)X[(int64_t)r * N + n];
                xx += x[r] * x[r];
            }
        } else {
            for (int r = 0; r < d; ++r) {
                x[r] = (double)X[(int64_t)r * N + n];
                xx += x[r] * x[r];
            }
        }
        const float cl = closest ? closest[n] : 0.0f;
#pragma unroll
        for (int j = 0; j < kPpMaxTrials; ++j) {
            if (j < n_trials) {
                double dot = 0.0;
                if constexpr (DIM != 0) {
#pragma unroll
                    for (int r = 0; r < DM; ++r) dot += cen[j][r] * x[r];
                } else {
                    for (int r = 0; r < d; ++r) dot += cen[j][r] * x[r];
                }
                const double dd = (-2.0 * dot + cen[j][d]) + xx;
                float f = (float)dd;
                f = f > 0.0f ? f : 0.0f;          // np.maximum(distances, 0)
                if (closest) f = cl < f ? cl : f;  // np.minimum(closest_dist_sq, distance_to_candidates)
                D[(int64_t)j * N + n] = f;
                sums[j] += (double)f;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < kPpMaxTrials; ++j) {
        if (j < n_trials) {
            const double s = wave_sum(sums[j]);
            if ((threadIdx.x & (kWave - 1)) == 0) part[j][threadIdx.x / kWave] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < n_trials) {
        double t = 0.0;
        for (int w = 0; w < kPpThreads / kWave; ++w) t += part[threadIdx.x][w];
        partials[(int64_t)threadIdx.x * n_blocks + blockIdx.x] = t;
    }
}

// One workgroup: potentials (fixed-order fp64 fold, rounded to float32), np.argmin (first minimum) -> the new centre.
__global__ __launch_bounds__(kPpThreads) void kpp_select_kernel(const double *__restrict__ partials, int64_t n_blocks,
                                                                 int n_trials, const int64_t *__restrict__ cand,
                                                                 const float *__restrict__ X, int64_t N, int d, int K, int c,
                                                                 float *__restrict__ pot, int *__restrict__ best,
                                                                 float *__restrict__ centers, int64_t *__restrict__ indices,
                                                                 PpBatch bt) {
    {
        const int64_t off = (int64_t)blockIdx.y * bt.ws;
        partials = pp_shift(partials, off), cand = pp_shift(cand, off), pot = pp_shift(pot, off), best = pp_shift(best, off);
        centers += (int64_t)blockIdx.y * bt.centers;
        indices += (int64_t)blockIdx.y * bt.indices;
    }
    __shared__ float pots[kPpMaxTrials];
    __shared__ int best_s;
    const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    for (int j = wave; j < n_trials; j += kPpThreads / kWave) {
        double s = 0.0;
        for (int64_t b = lane; b < n_blocks; b += kWave) s += partials[(int64_t)j * n_blocks + b];
        s = wave_sum(s);
        if (lane == 0) pots[j] = (float)s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int b = 0;
        for (int j = 1; j < n_trials; ++j)
            if (pots[j] < pots[b]) b = j;
        best_s = b;
        best[0] = b;
        pot[0] = pots[b];
        indices[c] = cand[b];
    }
    __syncthreads();
    if (threadIdx.x < d) centers[threadIdx.x * K + c] = X[(int64_t)threadIdx.x * N + cand[best_s]];
}

struct PpWorkspace {
    float *D;           // n_trials x N
    float *closest;     // N
    double *blocksums;  // nb
    double *prefix;     // nb
    double *partials;   // n_trials x nb
    int64_t *cand;      // kPpMaxTrials
    float *pot;         // 1
    int *best;          // 1
    size_t bytes;
};

static PpWorkspace pp_carve(void *ws, int64_t N, int n_trials) {
    const int64_t nb = ceil_div(N > 0 ? N : 1, (int64_t)kPpBlock);
    char *p = reinterpret_cast<char *>(ws);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *q = p ? p + off : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        return q;
    };
    PpWorkspace w;
    w.D = reinterpret_cast<float *>(take(sizeof(float) * (size_t)n_trials * (size_t)N));
    w.closest = reinterpret_cast<float *>(take(sizeof(float) * (size_t)N));
    w.blocksums = reinterpret_cast<double *>(take(sizeof(double) * (size_t)nb));
    w.prefix = reinterpret_cast<double *>(take(sizeof(double) * (size_t)nb));
    w.partials = reinterpret_cast<double *>(take(sizeof(double) * (size_t)n_trials * (size_t)nb));
    w.cand = reinterpret_cast<int64_t *>(take(sizeof(int64_t) * kPpMaxTrials));
    w.pot = reinterpret_cast<float *>(take(sizeof(float)));
    w.best = reinterpret_cast<int *>(take(sizeof(int)));
    w.bytes = off;
    return w;
}

}  // namespace et

using namespace et;

extern "C" int et_center_columns(float *X, int64_t N, int d, float rel_tol, float *mean, float *tol,
                                 void *workspace, size_t workspace_bytes, et_stream_t stream) {
    if (d < 1 || d > ET_KMEANS_MAX_D || N < 1 || !X || !mean || !tol) return ET_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < 2 * sizeof(float) * ET_KMEANS_MAX_D) return ET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float *var = reinterpret_cast<float *>(workspace) + ET_KMEANS_MAX_D;
    // sklearn's order (KMeans.fit): the tolerance from the variance of the data AS GIVEN (_check_params_vs_input ->
    // _tolerance: mean(np.var(X, axis=0)) * tol; np.var's own column mean is the same sequential fp32 sum / n as
    // X.mean(axis=0)), THEN X -= X.mean(axis=0)
    hipLaunchKernelGGL((colstats_kernel<0>), dim3(1), dim3(kPpThreads), colstats_lds_bytes(d), st, X, N, d, (const float *)nullptr,
                       mean);
    hipLaunchKernelGGL((colstats_kernel<1>), dim3(1), dim3(kPpThreads), colstats_lds_bytes(d), st, X, N, d, (const float *)mean,
                       var);
    const int64_t blocks = ceil_div(N * d, (int64_t)kPpThreads);
    hipLaunchKernelGGL(center_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(kPpThreads), 0, st, X, N, d,
                       (const float *)mean);
    hipLaunchKernelGGL(tolerance_kernel, dim3(1), dim3(64), 0, st, (const float *)var, d, rel_tol, tol);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" size_t et_kmeanspp_workspace_bytes(int64_t N, int d, int n_trials) {
    if (N < 1 || d < 1 || d > ET_KMEANS_MAX_D || n_trials < 1 || n_trials > kPpMaxTrials) return 0;
    return pp_carve(nullptr, N, n_trials).bytes;
}

static int pp_seed_launches(const float *X, int64_t N, int d, int K, int n_trials, const double *uniforms, float *centers,
                            int64_t *indices, const PpWorkspace &w, hipStream_t st, int64_t batch, size_t ws_stride);

extern "C" size_t et_kmeanspp_batch_workspace_bytes(int64_t N, int d, int n_trials, int64_t batch) {
    const size_t one = et_kmeanspp_workspace_bytes(N, d, n_trials);
    return one == 0 || batch < 1 ? 0 : one * (size_t)batch;
}

// `batch` seedings of the SAME points side by side: the y dimension of every launch (the n_init initialisations of the
// sklearn recipe: 4K - 1 launches for all of them together instead of per initialisation -- at dataset sizes a seeding is
// nothing but launch latency.  A one-launch form with a fence-free grid barrier between the phases was built and
// measured too: 1.3 ms against 0.8 ms, its ~15 dependent sc1 round trips to the memory side per centre cost more than
// the kernel boundaries they replace; profiles/r03g_calc_params_profile.txt)
extern "C" int et_kmeanspp_seed_batch(const float *X, int64_t N, int d, int K, int n_trials, const double *uniforms,
                                      int64_t batch, float *centers, int64_t *indices, void *workspace,
                                      size_t workspace_bytes, et_stream_t stream) {
    if (!X || !uniforms || !centers || !indices || N < 1 || d < 1 || d > ET_KMEANS_MAX_D || K < 1 ||
        K > ET_KMEANS_MAX_CLUSTERS || n_trials < 1 || n_trials > kPpMaxTrials || n_trials * d > kPpThreads || batch < 1 ||
        batch > 65535)
        return ET_ERR_INVALID_ARG;
    const size_t one = et_kmeanspp_workspace_bytes(N, d, n_trials);
    if (!workspace || workspace_bytes < one * (size_t)batch) return ET_ERR_WORKSPACE;
    const PpWorkspace w = pp_carve(workspace, N, n_trials);
    return pp_seed_launches(X, N, d, K, n_trials, uniforms, centers, indices, w, (hipStream_t)stream, batch, one);
}

// the 4K - 1 launches of `batch` seedings of the same points (grid.y = seeding)
static int pp_seed_launches(const float *X, int64_t N, int d, int K, int n_trials, const double *uniforms, float *centers,
                            int64_t *indices, const PpWorkspace &w, hipStream_t st, int64_t batch, size_t ws_stride) {
    const int64_t nb = ceil_div(N, (int64_t)kPpBlock);
    const unsigned B = (unsigned)batch;
    PpBatch bt;
    bt.ws = (int64_t)ws_stride;
    bt.uniforms = 1 + (int64_t)(K - 1) * n_trials;
    bt.centers = (int64_t)d * K;
    bt.indices = K;
    // first centre: index floor(u0 * N); closest = its distances (no min), potential = their sum
    hipLaunchKernelGGL(kpp_locate_kernel, dim3(1, B), dim3(kPpThreads), 0, st, (const float *)nullptr, N,
                       (const double *)nullptr, nb, w.prefix, uniforms, 1, 0, (const float *)nullptr, w.cand, bt);
    if (d == 6)
        hipLaunchKernelGGL(kpp_dist_kernel<6>, dim3((unsigned)nb, B), dim3(kPpThreads), 0, st, X, N, d, (const int64_t *)w.cand, 1,
                           (const float *)nullptr, w.D, w.partials, nb, bt);
    else
        hipLaunchKernelGGL(kpp_dist_kernel<0>, dim3((unsigned)nb, B), dim3(kPpThreads), 0, st, X, N, d, (const int64_t *)w.cand, 1,
                           (const float *)nullptr, w.D, w.partials, nb, bt);
    hipLaunchKernelGGL(kpp_select_kernel, dim3(1, B), dim3(kPpThreads), 0, st, (const double *)w.partials, nb, 1,
                       (const int64_t *)w.cand, X, N, d, K, 0, w.pot, w.best, centers, indices, bt);
    for (int c = 1; c < K; ++c) {
        hipLaunchKernelGGL(kpp_blocksum_kernel, dim3((unsigned)nb, B), dim3(kPpThreads), 0, st, (const float *)w.D,
                           (const int *)w.best, w.closest, N, w.blocksums, bt);
        hipLaunchKernelGGL(kpp_locate_kernel, dim3(1, B), dim3(kPpThreads), 0, st, (const float *)w.closest, N,
                           (const double *)w.blocksums, nb, w.prefix, uniforms + 1 + (size_t)(c - 1) * n_trials, n_trials, c,
                           (const float *)w.pot, w.cand, bt);
        if (d == 6)
            hipLaunchKernelGGL(kpp_dist_kernel<6>, dim3((unsigned)nb, B), dim3(kPpThreads), 0, st, X, N, d, (const int64_t *)w.cand,
                               n_trials, (const float *)w.closest, w.D, w.partials, nb, bt);
        else
            hipLaunchKernelGGL(kpp_dist_kernel<0>, dim3((unsigned)nb, B), dim3(kPpThreads), 0, st, X, N, d, (const int64_t *)w.cand,
                               n_trials, (const float *)w.closest, w.D, w.partials, nb, bt);
        hipLaunchKernelGGL(kpp_select_kernel, dim3(1, B), dim3(kPpThreads), 0, st, (const double *)w.partials, nb, n_trials,
                           (const int64_t *)w.cand, X, N, d, K, c, w.pot, w.best, centers, indices, bt);
    }
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_kmeanspp_seed(const float *X, int64_t N, int d, int K, int n_trials, const double *uniforms,
                                float *centers, int64_t *indices, void *workspace, size_t workspace_bytes,
                                et_stream_t stream) {
    return et_kmeanspp_seed_batch(X, N, d, K, n_trials, uniforms, 1, centers, indices, workspace, workspace_bytes, stream);
}
