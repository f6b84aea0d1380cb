// et_reforder_plain.inl -- part of csrc/et_kmeans_reforder.hip (ONE translation unit: this file is #included there, in order, and is
// not compiled on its own): the plain kernels (any shape): euc_sim, assignment, the inner cascade levels, farthest-first steps -- each in ATen's own summation order.
namespace et {
namespace reforder {

constexpr int kThreads = 256;
constexpr int kMaxD = ET_KMEANS_MAX_D;

__host__ __device__ inline int ceil_log2_aten(int64_t x) {  // c10::utils::CeilLog2
    if (x <= 2) return 1;
    int l = 0;
    for (int64_t v = x - 1; v > 0; v >>= 1) ++l;
    return l;
}
__host__ __device__ inline int level_power(int64_t size) {
    const int lp = ceil_log2_aten(size) / 4;
    return lp > 4 ? lp : 4;
}

// ATen multi_row_sum over `size` values v[0], v[stride], ...
__device__ inline float cascade_f32(const float *v, int stride, int size) {
    const int lp = level_power(size);
    const int step = 1 << lp, lmask = step - 1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int i = 0;
    while (i + step <= size) {
        for (int q = 0; q < step; ++q, ++i) acc[0] = acc[0] + v[i * stride];
        for (int lv = 1; lv < 4; ++lv) {
            acc[lv] = acc[lv] + acc[lv - 1];
            acc[lv - 1] = 0.f;
            if ((i & (lmask << (lv * lp))) != 0) break;
        }
    }
    for (; i < size; ++i) acc[0] = acc[0] + v[i * stride];
    for (int lv = 1; lv < 4; ++lv) acc[0] = acc[0] + acc[lv];
    return acc[0];
}
// ATen row_sum: 4 interleaved lanes, leftovers onto lane 0, lanes combined in order
__device__ inline float row_sum_f32(const float *v, int size) {
    const int s4 = size / 4;
    float lane[4];
    for (int k = 0; k < 4; ++k) lane[k] = cascade_f32(v + k, 4, s4);
    for (int i = s4 * 4; i < size; ++i) lane[0] = lane[0] + v[i];
    for (int k = 1; k < 4; ++k) lane[0] = lane[0] + lane[k];
    return lane[0];
}
// which order column `pos` of `count` gets in x.pow(2).sum(dim=-2)
__host__ __device__ inline bool column_is_sequential(int64_t pos, int64_t count) {
    return count < 8 ? pos < count / 4 * 4 : pos < count / 32 * 32;
}
// ATen vectorized_inner_sum over a contiguous array (kmeans.py:50)
__device__ inline float inner_sum_f32(const float *v, int size) {
    if (size < 8) return row_sum_f32(v, size);  // less than one vector: the scalar kernel's row_sum
    const int nv = size / 8;
    float lanes[8];
    for (int l = 0; l < 8; ++l) {
        const int s4 = nv / 4;
        float slot[4];
        for (int k = 0; k < 4; ++k) slot[k] = cascade_f32(v + 8 * k + l, 32, s4);
        for (int i = s4 * 4; i < nv; ++i) slot[0] = slot[0] + v[8 * i + l];
        for (int k = 1; k < 4; ++k) slot[0] = slot[0] + slot[k];
        lanes[l] = slot[0];
    }
    float acc = 0.f;
    for (int i = nv * 8; i < size; ++i) acc = acc + v[i];
    for (int l = 0; l < 8; ++l) acc = acc + lanes[l];
    return acc;
}

__device__ inline float sqnorm_at(const float *sq, int d, int64_t pos, int64_t count) {
    if (count == 1 && d >= 8) return inner_sum_f32(sq, d);  // one column of >= 8 rows: a contiguous reduction for ATen
    return column_is_sequential(pos, count) ? cascade_f32(sq, 1, d) : row_sum_f32(sq, d);
}

// torch.max (kmeans.py:156): NaN beats everything, first index wins
__device__ inline bool gt_nanmax(float cand, float best) { return (cand > best) || (isnan(cand) && !isnan(best)); }
// torch.argmin (kmeans.py:97): NaN is the smallest, first index wins.  Is (v1, i1) ahead of (v2, i2)?
__device__ inline bool argmin_ahead(float v1, long long i1, float v2, long long i2) {
    const bool n1 = isnan(v1), n2 = isnan(v2);
    if (n1 != n2) return n1;
    if (!n1 && v1 != v2) return v1 < v2;
    return i1 < i2;
}

// |c_j|^2 of the `count` centroid columns currently in play, into LDS
__device__ inline void stage_centroid_norms(const float *cen, int d, int K, int count, float *sC, float *sBn) {
    for (int e = threadIdx.x; e < d * count; e += blockDim.x) sC[e] = cen[(e / count) * K + (e % count)];
    __syncthreads();
    for (int j = threadIdx.x; j < count; j += blockDim.x) {
        float sq[kMaxD];
        for (int i = 0; i < d; ++i) {
            const float v = sC[i * count + j];
            sq[i] = v * v;
        }
        sBn[j] = sqnorm_at(sq, d, j, count);
    }
    __syncthreads();
}

// max_j euc_sim(x_n, c_j) over `count` centroids and its arg-max
__device__ inline void best_of(const float *X, int64_t N, int d, int64_t n, const float *sC, const float *sBn, int count,
                               float &best, int &lb) {
    float x[kMaxD], sq[kMaxD];
    for (int i = 0; i < d; ++i) {
        x[i] = X[(int64_t)i * N + n];
        sq[i] = x[i] * x[i];
    }
    const float an = sqnorm_at(sq, d, n, N);
    best = 0.f;
    lb = 0;
    for (int j = 0; j < count; ++j) {
        float y = 0.f;
        for (int i = 0; i < d; ++i) y = fmaf(x[i], sC[i * count + j], y);
        y = y * 2.0f;
        y = y - an;
        y = y - sBn[j];
        if (j == 0 || gt_nanmax(y, best)) {
            best = y;
            lb = j;
        }
    }
}

// ---- kmeans.py:143-158: labels, maxsims, per-cluster counts ----
__global__ __launch_bounds__(kThreads) void reforder_assign_kernel(const float *__restrict__ X, int64_t N, int d, int K,
                                                                   const float *__restrict__ cen, uint8_t *__restrict__ labels,
                                                                   float *__restrict__ maxsims,
                                                                   unsigned long long *__restrict__ counts) {
    extern __shared__ float smem[];
    float *sC = smem, *sBn = smem + d * K;
    __shared__ unsigned sCnt[256];
    for (int j = threadIdx.x; j < 256; j += blockDim.x) sCnt[j] = 0u;
    stage_centroid_norms(cen, d, K, K, sC, sBn);
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        float best;
        int lb;
        best_of(X, N, d, n, sC, sBn, K, best, lb);
        labels[n] = (uint8_t)lb;
        maxsims[n] = best;
        atomicAdd(&sCnt[lb], 1u);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < K; j += blockDim.x)
        if (sCnt[j]) atomicAdd(&counts[j], (unsigned long long)sCnt[j]);
}

// ---- kmeans.py:180-182, inner two cascade levels ----
// work item (g, lane, column): group g = L consecutive level-0 chunks of L lane terms each (the last group may hold
// fewer full chunks); S1[g][lane][column] = the level-1 accumulator after those chunks
__global__ __launch_bounds__(kThreads) void reforder_group_kernel(const float *__restrict__ X, int64_t N, int d, int K,
                                                                  const uint8_t *__restrict__ labels, int lp, int64_t n_groups,
                                                                  int64_t full_chunks, float *__restrict__ S1) {
    const int dk = d * K;
    const int64_t total = n_groups * 4 * dk;
    const int64_t L = (int64_t)1 << lp;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(w % dk);
        const int lane = (int)((w / dk) % 4);
        const int64_t g = w / (4 * dk);
        const int j = e % K;
        const float *x = X + (int64_t)(e / K) * N;
        float acc1 = 0.f;
        for (int64_t c = g * L; c < (g + 1) * L && c < full_chunks; ++c) {
            float acc0 = 0.f;
            for (int64_t r = c * L; r < (c + 1) * L; ++r) {
                const int64_t n = 4 * r + lane;
                if (labels[n] == j) acc0 = acc0 + x[n];
            }
            acc1 = acc1 + acc0;
        }
        S1[w] = acc1;
    }
}

// outer two levels, leftovers, lane combination -> sums (d, K)
__global__ __launch_bounds__(kThreads) void reforder_finish_kernel(const float *__restrict__ X, int64_t N, int d, int K,
                                                                   const uint8_t *__restrict__ labels, int lp,
                                                                   int64_t full_chunks, const float *__restrict__ S1,
                                                                   float *__restrict__ lanes, float *__restrict__ sums) {
    const int dk = d * K;
    const int64_t L = (int64_t)1 << lp;
    const int64_t size = N / 4, full_groups = full_chunks / L;
    for (int w = threadIdx.x; w < 4 * dk; w += blockDim.x) {
        const int e = w % dk, lane = w / dk;
        const int j = e % K;
        const float *x = X + (int64_t)(e / K) * N;
        float acc2 = 0.f, acc3 = 0.f;
        for (int64_t g = 0; g < full_groups; ++g) {
            acc2 = acc2 + S1[(g * 4 + lane) * dk + e];
            if ((g + 1) % L == 0) {
                acc3 = acc3 + acc2;
                acc2 = 0.f;
            }
        }
        const float acc1 = full_chunks % L ? S1[(full_groups * 4 + lane) * dk + e] : 0.f;
        float acc0 = 0.f;
        for (int64_t r = full_chunks * L; r < size; ++r) {
            const int64_t n = 4 * r + lane;
            if (labels[n] == j) acc0 = acc0 + x[n];
        }
        lanes[w] = ((acc0 + acc1) + acc2) + acc3;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < dk; e += blockDim.x) {
        const int j = e % K;
        const float *x = X + (int64_t)(e / K) * N;
        float p = lanes[e];
        for (int64_t n = size * 4; n < N; ++n)
            if (labels[n] == j) p = p + x[n];
        for (int lane = 1; lane < 4; ++lane) p = p + lanes[lane * dk + e];
        sums[e] = p;
    }
}

// deterministic fp64 partial sums of the maxsims (the inertia is only printed by the reference, kmeans.py:236)
__global__ __launch_bounds__(kThreads) void reforder_inertia_kernel(const float *__restrict__ maxsims, int64_t N,
                                                                    double *__restrict__ partial) {
    __shared__ double sW[kThreads];
    double s = 0.0;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x)
        s = s + (double)maxsims[n];
    sW[threadIdx.x] = s;
    __syncthreads();
    for (int o = kThreads / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sW[threadIdx.x] = sW[threadIdx.x] + sW[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sW[0];
}

// kmeans.py:231-239: centroids = sums / counts, error, inertia, stop flag.  One workgroup.
__global__ __launch_bounds__(kThreads) void reforder_update_kernel(et_kmeans_state *state, const float *__restrict__ sums,
                                                                   unsigned long long *__restrict__ counts,
                                                                   const double *__restrict__ partial, int n_partial,
                                                                   int64_t N, int d, int K, float tol, float *__restrict__ cen,
                                                                   float *__restrict__ trace) {
    extern __shared__ float smem[];
    float *sSq = smem;
    const int dk = d * K;
    for (int e = threadIdx.x; e < dk; e += blockDim.x) {
        const float c = sums[e] / (float)(long long)counts[e % K];  // 0/0 = NaN for an empty cluster (kmeans.py:182)
        const float diff = cen[e] - c;
        sSq[e] = diff * diff;
        cen[e] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float error = inner_sum_f32(sSq, dk);
        double s = 0.0;
        for (int b = 0; b < n_partial; ++b) s = s + partial[b];
        const float inertia = (float)(-(s / (double)N));
        const int64_t it = state->iter;
        if (trace) {
            trace[2 * it] = error;
            trace[2 * it + 1] = inertia;
        }
        state->error = (double)error;
        state->inertia = (double)inertia;
        state->iter = it + 1;
        state->done = (error <= tol) ? 1 : 0;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < K; j += blockDim.x) counts[j] = 0ull;  // for the next assignment
}

// ---- kmeans.py:88-112 farthest-first: step with `count` centroids known ----
struct Cand {
    float v;
    int pad;
    long long idx;
};
__global__ __launch_bounds__(kThreads) void reforder_init_step_kernel(const float *__restrict__ X, int64_t N, int d, int K,
                                                                      int count, const float *__restrict__ C0,
                                                                      Cand *__restrict__ cands) {
    extern __shared__ float smem[];
    float *sC = smem, *sBn = smem + d * count;
    stage_centroid_norms(C0, d, K, count, sC, sBn);
    float bv = 0.f;
    long long bi = -1;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        float best;
        int lb;
        best_of(X, N, d, n, sC, sBn, count, best, lb);
        if (bi < 0 || argmin_ahead(best, n, bv, bi)) {
            bv = best;
            bi = n;
        }
    }
    __shared__ float sV[kThreads];
    __shared__ long long sI[kThreads];
    sV[threadIdx.x] = bv;
    sI[threadIdx.x] = bi;
    __syncthreads();
    for (int o = kThreads / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const float v2 = sV[threadIdx.x + o];
            const long long i2 = sI[threadIdx.x + o];
            if (i2 >= 0 && (sI[threadIdx.x] < 0 || argmin_ahead(v2, i2, sV[threadIdx.x], sI[threadIdx.x]))) {
                sV[threadIdx.x] = v2;
                sI[threadIdx.x] = i2;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        cands[blockIdx.x].v = sV[0];
        cands[blockIdx.x].idx = sI[0];
    }
}
// ---- the same step, incrementally (d < 8, K <= 32): the reference re-evaluates euc_sim against ALL `count` current
// centroids, but which of the two norm orders a centroid's |b|^2 takes depends on (column, count) in a way that leaves only two
// regimes for count <= 31: every column in the 4-lane order ("R"), except count = 4 .. 7, where columns 0 .. 3 are summed in
// sequence ("S") (column_is_sequential).  So a running maximum over the R-order similarities (bestR: all centroids; bestR4:
// centroids >= 4) is exact -- max is order independent, NaN sticky -- and a step evaluates ONE new centroid per point (plus,
// in the four steps count = 4 .. 7, the S-order similarities of centroids 0 .. 3) instead of `count`:
//   count in 1..3, 8..31:  value = bestR;      count in 4..7:  value = max(max_{j<4} sim_S(x, c_j), bestR4)
// 9.0 -> ~1.5 ms for the 19 steps at 1e7 points; the same picks bit for bit (tests: every G7c / G7d case, odd shapes).
template <int D>  // D = 6: the coordinates in registers; 0: any d < 8 (run-time loops)
__global__ __launch_bounds__(kThreads) void reforder_init_step_inc_kernel(const float *__restrict__ X, int64_t N, int d_rt, int K,
                                                                          int count, const float *__restrict__ C0,
                                                                          float *__restrict__ bestR, float *__restrict__ bestR4,
                                                                          uint8_t *__restrict__ nearest, unsigned *__restrict__ max_abs_bits,
                                                                          int skip_ok, Cand *__restrict__ cands,
                                                                          const Cand *__restrict__ prev_cands, int n_prev,
                                                                          float *__restrict__ C0_rw) {
    constexpr int DM = D ? D : 8;
    const int d = D ? D : d_rt;
    __shared__ float sV[kThreads];
    __shared__ long long sI[kThreads];
    __shared__ float sNew[DM + 1];      // the newest centroid (column count - 1) and its R-order norm
    __shared__ float sS[4 * (DM + 1)];  // count in 4..7: centroids 0..3 and their S-order norms
    __shared__ float sDelta[ET_KMEANS_MAX_CLUSTERS];  // lower bounds of ||c_new - c_j||^2, j < count - 1
    __shared__ unsigned sMabs;
    const bool window = count >= 4 && count <= 7;
    // Outside the window the step's value IS bestR, and a point whose nearest centroid c_l (the arg-max behind bestR) is
    // closer than half the distance from c_l to the new centroid cannot get a larger similarity from the new one -- the test
    // of csrc/et_kmeans.hip's farthest-first (init_step_body: ||c_new - c_l||^2 >= 4 (E - b), E >= twice the rounding error
    // of the similarity formula in ANY summation order of the norms), on the reference-order values: such a point costs
    // 5 bytes (bestR, nearest) instead of 28, and its value is bit for bit what the full evaluation would leave.
    // (skip_ok: only for big shards -- below ~2e6 points a step is two dependent round trips instead of one and nothing else)
    const bool can_skip = count >= 2 && !window && skip_ok != 0;
    if (threadIdx.x == 0) sMabs = 0u;
    // Centroid count - 1 is the winner of the PREVIOUS step's workgroup candidates: every workgroup derives it itself (the same
    // reduction everywhere; workgroup 0 also stores it into C0) -- two short round trips in the prologue instead of a pick
    // launch between two steps (19 launches and their boundaries per seeding).  prev_cands == nullptr: it is in C0 already.
    if (prev_cands) {
        float pv = 0.f;
        long long pi = -1;
        for (int b = threadIdx.x; b < n_prev; b += kThreads) {
            const float v = prev_cands[b].v;
            const long long i = prev_cands[b].idx;
            if (i >= 0 && (pi < 0 || argmin_ahead(v, i, pv, pi))) {
                pv = v;
                pi = i;
            }
        }
        sV[threadIdx.x] = pv;
        sI[threadIdx.x] = pi;
        __syncthreads();
        for (int o = kThreads / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) {
                const float v2 = sV[threadIdx.x + o];
                const long long i2 = sI[threadIdx.x + o];
                if (i2 >= 0 && (sI[threadIdx.x] < 0 || argmin_ahead(v2, i2, sV[threadIdx.x], sI[threadIdx.x]))) {
                    sV[threadIdx.x] = v2;
                    sI[threadIdx.x] = i2;
                }
            }
            __syncthreads();
        }
        if ((int)threadIdx.x < d) {
            const float v = X[(int64_t)threadIdx.x * N + sI[0]];
            sNew[threadIdx.x] = v;
            if (blockIdx.x == 0) C0_rw[threadIdx.x * K + (count - 1)] = v;
        }
        __syncthreads();
    } else if ((int)threadIdx.x < d) {
        sNew[threadIdx.x] = C0[threadIdx.x * K + (count - 1)];
    }
    if (!prev_cands) __syncthreads();
    if (can_skip && (int)threadIdx.x >= 128 && (int)threadIdx.x < 128 + count - 1) {
        const int j = (int)threadIdx.x - 128;
        double s2 = 0.0;
        for (int i = 0; i < d; ++i) {
            const double t = (double)sNew[i] - (double)C0[i * K + j];
            s2 += t * t;
        }
        sDelta[j] = (float)(s2 * (1.0 - 4e-6)) * (1.0f - 1e-6f);
    }
    if (threadIdx.x == 0) {
        float sq[kMaxD];
        for (int i = 0; i < d; ++i) sq[i] = sNew[i] * sNew[i];
        sNew[DM] = row_sum_f32(sq, d);
    }
    if (window && threadIdx.x >= 64 && threadIdx.x < 68) {
        const int j = threadIdx.x - 64;
        float sq[kMaxD];
        for (int i = 0; i < d; ++i) {
            const float v = j == count - 1 ? sNew[i] : C0[i * K + j];  // (count = 4: column 3 is being stored by workgroup 0 right now)
            sS[j * (DM + 1) + i] = v;
            sq[i] = v * v;
        }
        sS[j * (DM + 1) + DM] = cascade_f32(sq, 1, d);
    }
    __syncthreads();
    float bv = 0.f;
    long long bi = -1;
    float E = __int_as_float(0x7f800000);  // (unknown: nothing is skipped)
    if (can_skip) {
        const float R = 2.0f * sqrtf((float)d) * __uint_as_float(*max_abs_bits) * 1.0001f;  // every centroid is a point
        E = R * R * 1.9073486328125e-6f;                                                   // 2^-19 (|x| + |c|)^2
        if (!(E <= 3.0e38f)) E = __int_as_float(0x7f800000);
    }
    float mabs = 0.f;
    const int64_t seq_cols = N < 8 ? N / 4 * 4 : N / 32 * 32;  // column_is_sequential(n, N)
    // one point: `known` = its bestR is in b already and the skip test has been made (passed: skip)
    auto visit = [&](int64_t n, bool known, float b, bool skip) {
        if (skip) {
            if (bi < 0 || argmin_ahead(b, n, bv, bi)) {
                bv = b;
                bi = n;
            }
            return;
        }
        float x[DM];
#pragma unroll
        for (int i = 0; i < DM; ++i) x[i] = i < d ? X[(int64_t)i * N + n] : 0.f;
        if (count == 1) {
#pragma unroll
            for (int i = 0; i < DM; ++i) mabs = fmaxf(mabs, fabsf(x[i]));  // (a NaN is ignored here and never skipped later)
        }
        float an;
        if (n < seq_cols) {  // rows in sequence (0 + s0 = s0)
            an = x[0] * x[0];
#pragma unroll
            for (int i = 1; i < DM; ++i)
                if (i < d) an = an + x[i] * x[i];
        } else {
            float sq[kMaxD];
            for (int i = 0; i < d; ++i) sq[i] = x[i] * x[i];
            an = sqnorm_at(sq, d, n, N);
        }
        auto sim = [&](const float *c) {
            float y = 0.f;
#pragma unroll
            for (int i = 0; i < DM; ++i)
                if (i < d) y = fmaf(x[i], c[i], y);
            y = y * 2.0f;
            y = y - an;
            y = y - c[DM];
            return y;
        };
        const float yn = sim(sNew);
        float r = count == 1 ? yn : (known ? b : bestR[n]);
        const bool took = count == 1 || gt_nanmax(yn, r);
        if (took) {
            r = yn;
            bestR[n] = r;
            nearest[n] = (uint8_t)(count - 1);
        }
        float value = r;
        if (count >= 5 && count <= 7) {  // (bestR4 is only ever read inside the window)
            float r4 = count == 5 ? yn : bestR4[n];
            if (count > 5 && gt_nanmax(yn, r4)) r4 = yn;
            bestR4[n] = r4;
            value = r4;
        }
        if (window) {
            float v = sim(sS);
            for (int j = 1; j < 4; ++j) {
                const float y = sim(sS + j * (DM + 1));
                if (gt_nanmax(y, v)) v = y;
            }
            if (count >= 5 && gt_nanmax(value, v)) v = value;
            value = v;
        }
        if (bi < 0 || argmin_ahead(value, n, bv, bi)) {
            bv = value;
            bi = n;
        }
    };
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gstride = (int64_t)gridDim.x * blockDim.x;
    if (can_skip) {
        // four points per lane through one 16-byte load of bestR and one 4-byte load of nearest (one point per lane and trip
        // was a chain of ~38 dependent round trips per thread at 1e7 points: 48 us per step whatever it skipped)
        const int64_t n4 = N / 4;
        for (int64_t g = gtid; g < n4; g += gstride) {
            const float4 b4 = reinterpret_cast<const float4 *>(bestR)[g];
            const unsigned l4 = reinterpret_cast<const unsigned *>(nearest)[g];
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float w = E - bb[v];  // (a NaN or +inf anywhere makes a comparison false: full evaluation)
                const bool skip = w >= 0.0f && sDelta[(l4 >> (8 * v)) & 0xffu] >= 4.0001f * w;
                visit(4 * g + v, true, bb[v], skip);
            }
        }
        for (int64_t n = 4 * n4 + gtid; n < N; n += gstride) {
            const float b = bestR[n];
            const float w = E - b;
            visit(n, true, b, w >= 0.0f && sDelta[nearest[n]] >= 4.0001f * w);
        }
    } else {
        for (int64_t n = gtid; n < N; n += gstride) visit(n, false, 0.f, false);
    }
    sV[threadIdx.x] = bv;
    sI[threadIdx.x] = bi;
    if (count == 1 && mabs > 0.f) atomicMax(&sMabs, __float_as_uint(mabs));  // (non-negative floats order like their bits)
    __syncthreads();
    for (int o = kThreads / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const float v2 = sV[threadIdx.x + o];
            const long long i2 = sI[threadIdx.x + o];
            if (i2 >= 0 && (sI[threadIdx.x] < 0 || argmin_ahead(v2, i2, sV[threadIdx.x], sI[threadIdx.x]))) {
                sV[threadIdx.x] = v2;
                sI[threadIdx.x] = i2;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        cands[blockIdx.x].v = sV[0];
        cands[blockIdx.x].idx = sI[0];
        if (count == 1 && sMabs) atomicMax(max_abs_bits, sMabs);
    }
}
// the winner of the blocks' candidates becomes column `col`; col = 0: the given first index
__global__ __launch_bounds__(kThreads) void reforder_init_pick_kernel(const float *__restrict__ X, int64_t N, int d, int K, int col,
                                                                      const Cand *__restrict__ cands, int n_cands,
                                                                      int64_t first_index, float *__restrict__ C0) {
    __shared__ float sV[kThreads];
    __shared__ long long sI[kThreads];
    // (all candidates requested side by side: one thread walking up to 1024 of them was ~200 us of every step)
    float bv = 0.f;
    long long bi = col > 0 ? -1 : first_index;
    if (col > 0)
        for (int b = threadIdx.x; b < n_cands; b += kThreads) {
            const float v = cands[b].v;
            const long long i = cands[b].idx;
            if (i >= 0 && (bi < 0 || argmin_ahead(v, i, bv, bi))) {
                bv = v;
                bi = i;
            }
        }
    sV[threadIdx.x] = bv;
    sI[threadIdx.x] = bi;
    __syncthreads();
    for (int o = kThreads / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const float v2 = sV[threadIdx.x + o];
            const long long i2 = sI[threadIdx.x + o];
            if (i2 >= 0 && (sI[threadIdx.x] < 0 || argmin_ahead(v2, i2, sV[threadIdx.x], sI[threadIdx.x]))) {
                sV[threadIdx.x] = v2;
                sI[threadIdx.x] = i2;
            }
        }
        __syncthreads();
    }
    const long long idx = sI[0];
    for (int i = threadIdx.x; i < d; i += kThreads) C0[i * K + col] = X[(int64_t)i * N + idx];
}

// kmeans.py:59-76 with both norms in torch's order: a (d,m), b (d,n) -> y (m,n)
__global__ __launch_bounds__(kThreads) void reforder_euc_sim_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                                    int d, int64_t m, int64_t n, float *__restrict__ y) {
    const int64_t total = m * n;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = w / n, j = w % n;
        float sa[kMaxD], sb[kMaxD];
        float dot = 0.f;
        for (int t = 0; t < d; ++t) {
            const float av = a[(int64_t)t * m + i], bv = b[(int64_t)t * n + j];
            sa[t] = av * av;
            sb[t] = bv * bv;
            dot = fmaf(av, bv, dot);
        }
        float v = dot * 2.0f;
        v = v - sqnorm_at(sa, d, i, m);
        v = v - sqnorm_at(sb, d, j, n);
        y[w] = v;
    }
}

struct Workspace {
    et_kmeans_state *state;
    uint8_t *labels_u8;
    float *maxsims;
    float *best4;  // farthest-first, incremental form: running maximum over centroids >= 4
    unsigned long long *counts;
    float *sums;
    float *lanes;
    double *partial;
    Cand *cands;
    float *S1;
    size_t bytes;
};
constexpr int kMaxBlocks = 1024;
static size_t up(size_t v) { return (v + 255) / 256 * 256; }
static Workspace carve(void *base, int64_t N, int d, int K) {
    Workspace w;
    unsigned char *p = (unsigned char *)base;
    size_t off = 0;
    const size_t dk = (size_t)d * K;
    w.state = (et_kmeans_state *)(p + off);
    off = up(off + sizeof(et_kmeans_state));
    w.labels_u8 = p + off;
    off = up(off + (size_t)N + 4);
    w.maxsims = (float *)(p + off);
    off = up(off + sizeof(float) * (size_t)N);
    w.best4 = (float *)(p + off);
    off = up(off + sizeof(float) * (size_t)N);
    w.counts = (unsigned long long *)(p + off);
    off = up(off + sizeof(unsigned long long) * 256);
    w.sums = (float *)(p + off);
    off = up(off + sizeof(float) * dk);
    w.lanes = (float *)(p + off);
    off = up(off + sizeof(float) * 4 * dk);
    w.partial = (double *)(p + off);
    off = up(off + sizeof(double) * kMaxBlocks);
    w.cands = (Cand *)(p + off);  // (two buffers: the incremental farthest-first reads one step's while it writes the next's)
    off = up(off + sizeof(Cand) * 2 * kMaxBlocks);
    const int lp = level_power(N / 4);
    const int64_t L = (int64_t)1 << lp;
    const int64_t groups = (N / 4 / L + L - 1) / L + 1;
    w.S1 = (float *)(p + off);
    off = up(off + sizeof(float) * (size_t)groups * 4 * dk);
    w.bytes = off;
    return w;
}
static bool dims_ok(int d, int K) { return d >= 1 && d <= ET_KMEANS_MAX_D && K >= 1 && K <= ET_KMEANS_MAX_CLUSTERS; }
static int grid_for(int64_t items) {
    const int64_t b = (items + kThreads - 1) / kThreads;
    return (int)(b < 1 ? 1 : (b > kMaxBlocks ? kMaxBlocks : b));
}
