// et_reforder_sharded.inl -- part of csrc/et_kmeans_reforder.hip (ONE translation unit: this file is #included there, in order, and is
// not compiled on its own): the reference-order iteration over shards: level-2 rows gathered over ranks, finish kernel, fast_fit host loop.
// =====================================================================================================================
// The reference-order iteration over SHARDS (one process per GPU; not in the reference).  The order of a cascade sum is a
// property of the whole array, but its tree is made of index ranges: with every shard boundary on a multiple of a level-2
// block (4 L^3 points, L from the TOTAL number of points) a rank owns whole blocks, runs levels 0 .. 2 of its own rows
// exactly as above, and what has to travel is one row of d K sums (+ K counts) per block -- 2 KB per 16 384 points at
// L = 16, per 1 048 576 at L = 64 -- plus the last rank's leftovers: ONE all-gather per iteration; then every rank runs
// the same sequential level 3 over the ranks' rows in rank order (= global block order), the lane combination, the update
// and the stop flag: identical centroids everywhere without a broadcast, and the same bits as the single-GPU fit.
// Record of a rank (16-byte words): rows[max_rows][d K + 8] | T1[d K] | T0[d K] | leftover labels | leftover coordinates
// (3 points x 6, 5 words) | tail counts (8) | similarity sum (fp64 in one word).
// =====================================================================================================================
struct ShardRec {
    int max_rows, rowlen, dk;
    __host__ __device__ int t1() const { return max_rows * rowlen; }
    __host__ __device__ int t0() const { return t1() + dk; }
    __host__ __device__ int lab() const { return t0() + dk; }
    __host__ __device__ int coords() const { return lab() + 1; }
    __host__ __device__ int tailcnt() const { return coords() + 5; }
    __host__ __device__ int sin() const { return tailcnt() + kFMaxK / 4; }
    __host__ __device__ int words() const { return sin() + 1; }
};

// levels 2 of this rank's blocks -> its record (plain stores: the all-gather follows the kernel); workgroup 0 adds the
// tail's rows, the leftover points, the tail's counts and the rank's similarity sum
__global__ __launch_bounds__(kUThreads) void reforder_level2_sharded_kernel(const Args a, ShardRec rec, int rows_local,
                                                                           float4 *__restrict__ send, int rows_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, dk = kD * K;
    unsigned char *ws = a.ws;
    const et_kmeans_state *state = at<et_kmeans_state>(ws, a.lay.state);
    if (state->done) return;
    const Geo &geo = a.geo;
    const int lp = geo.lp, L = 1 << lp;
    const float4 *S1 = at<const float4>(ws, a.lay.S1);
    const uint4 *cnt4 = at<const uint4>(ws, a.lay.cnt);
    float4 *sRows = reinterpret_cast<float4 *>(smem);
    const int blk = (int)blockIdx.x, rowlen = rec.rowlen;
    if (blk < rows_local) {
        const int64_t g0 = (int64_t)blk << lp;
        const int ng = (int)((geo.G - g0) < L ? (geo.G - g0) : L);
        float4 a2 = make_float4(0.f, 0.f, 0.f, 0.f);
        uint4 c2 = make_uint4(0u, 0u, 0u, 0u);
        for (int r0 = 0; r0 < ng; r0 += rows_cap) {
            const int nr = ng - r0 < rows_cap ? ng - r0 : rows_cap;
            const float4 *src = S1 + (g0 + r0) * dk;
            const uint4 *csrc = cnt4 + (g0 + r0) * (kFMaxK / 4);
            for (int r8 = 0; r8 < nr; r8 += 16) {
                if (tid < rowlen) {
                    float4 v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int r = r8 + u < nr ? r8 + u : r8;
                        v[u] = tid < dk ? src[r * dk + tid] : __builtin_bit_cast(float4, csrc[r * (kFMaxK / 4) + (tid - dk)]);
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (r8 + u < nr) sRows[(r8 + u) * rowlen + tid] = v[u];
                }
            }
            __syncthreads();
            if (tid < dk) {
                for (int g = 0; g < nr; ++g) {
                    const float4 v = sRows[g * rowlen + tid];
                    a2.x = a2.x + v.x;
                    a2.y = a2.y + v.y;
                    a2.z = a2.z + v.z;
                    a2.w = a2.w + v.w;
                }
            } else if (tid < rowlen) {
                for (int g = 0; g < nr; ++g) {
                    const uint4 v = __builtin_bit_cast(uint4, sRows[g * rowlen + tid]);
                    c2.x += v.x;
                    c2.y += v.y;
                    c2.z += v.z;
                    c2.w += v.w;
                }
            }
            __syncthreads();
        }
        if (tid < dk) send[(int64_t)blk * rowlen + tid] = a2;
        else if (tid < rowlen) send[(int64_t)blk * rowlen + tid] = __builtin_bit_cast(float4, c2);
    }
    if (blk != 0) return;
    const float4 *T = at<const float4>(ws, a.lay.T);
    const double *Sin = at<const double>(ws, a.lay.Sin);
    if (tid < dk) {
        send[rec.t1() + tid] = T[tid];
        send[rec.t0() + tid] = T[dk + tid];
    }
    if (tid == 0) send[rec.lab()] = T[2 * dk];
    if (tid < 5) {  // the N mod 4 points after the lanes' ranges: their coordinates travel with the record
        const int64_t N = geo.N, n0 = N / 4 * 4;
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = 4 * tid + u, pnt = e / kD, i = e % kD;
            v[u] = (e < 3 * kD && n0 + pnt < N) ? a.X[(int64_t)i * N + n0 + pnt] : 0.f;
        }
        send[rec.coords() + tid] = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (tid >= 64 && tid < 64 + kFMaxK / 4) send[rec.tailcnt() + (tid - 64)] = __builtin_bit_cast(float4, cnt4[geo.G * (kFMaxK / 4) + (tid - 64)]);
    __shared__ double sWsum[8];
    double part = 0.0;
    for (int64_t g = tid; g <= geo.G; g += kUThreads) part = part + Sin[g];
    part = wave_sum_f64(part);
    if (lane == 0) sWsum[wave] = part;
    __syncthreads();
    if (tid == 0) {
        double sum = sWsum[0];
        for (int w = 1; w < kUThreads / 64; ++w) sum = sum + sWsum[w];
        const unsigned long long b = (unsigned long long)__double_as_longlong(sum);
        send[rec.sin()] = make_float4(__uint_as_float((unsigned)b), __uint_as_float((unsigned)(b >> 32)), 0.f, 0.f);
    }
}

// every rank, identically: level 3 over the ranks' complete blocks in rank order, the tail rank's partial block / tail /
// leftovers, lane combination, new centroids (kmeans.py:180-182), error (ATen's inner sum), stop flag
__global__ __launch_bounds__(kUThreads) void reforder_finish_sharded_kernel(const Args a, ShardRec rec, int P, const int *__restrict__ rows_of,
                                                                           int tail_rank, int tail_full_rows, int64_t N_total,
                                                                           const float4 *__restrict__ table, int rows_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = (int)threadIdx.x;
    const int K = a.K, dk = kD * K, rowlen = rec.rowlen;
    et_kmeans_state *state = at<et_kmeans_state>(a.ws, a.lay.state);
    if (state->done) return;
    float *cen = at<float>(a.ws, a.lay.cen);
    __shared__ unsigned sCntTot[kFMaxK];
    __shared__ float sScr[40];
    float4 *sRows = reinterpret_cast<float4 *>(smem);
    float4 a3 = make_float4(0.f, 0.f, 0.f, 0.f), p2 = a3;
    unsigned ctot[4] = {0u, 0u, 0u, 0u};
    const int words = rec.words();
    for (int r = 0; r < P; ++r) {
        const float4 *rr = table + (int64_t)r * words;
        const int nrows = rows_of[r], nfull = r == tail_rank ? tail_full_rows : nrows;
        for (int r0 = 0; r0 < nrows; r0 += rows_cap) {
            const int nr = nrows - r0 < rows_cap ? nrows - r0 : rows_cap;
            for (int e = tid; e < nr * rowlen; e += kUThreads) sRows[e] = rr[r0 * rowlen + e];
            __syncthreads();
            if (tid < dk) {
                for (int b = 0; b < nr; ++b) {
                    const float4 v = sRows[b * rowlen + tid];
                    if (r0 + b < nfull) {
                        a3.x = a3.x + v.x;
                        a3.y = a3.y + v.y;
                        a3.z = a3.z + v.z;
                        a3.w = a3.w + v.w;
                    } else {
                        p2 = v;  // (the partial block: the tail rank's last row)
                    }
                }
            } else if (tid < rowlen) {
                for (int b = 0; b < nr; ++b) {
                    const uint4 v = __builtin_bit_cast(uint4, sRows[b * rowlen + tid]);
                    ctot[0] += v.x;
                    ctot[1] += v.y;
                    ctot[2] += v.z;
                    ctot[3] += v.w;
                }
            }
            __syncthreads();
        }
        if (tid >= dk && tid < rowlen) {
            const uint4 t = __builtin_bit_cast(uint4, rr[rec.tailcnt() + (tid - dk)]);
            ctot[0] += t.x;
            ctot[1] += t.y;
            ctot[2] += t.z;
            ctot[3] += t.w;
        }
    }
    if (tid >= dk && tid < rowlen) {
#pragma unroll
        for (int u = 0; u < 4; ++u) sCntTot[4 * (tid - dk) + u] = ctot[u];
    }
    __syncthreads();
    const float4 *last = table + (int64_t)tail_rank * words;  // (the rank that owns the end of the array)
    float *sSq = reinterpret_cast<float *>(smem);
    if (tid < dk) {
        const int j = tid % K, i = tid / K;
        const float4 p1 = last[rec.t1() + tid], p0 = last[rec.t0() + tid];
        const unsigned lw = __float_as_uint(last[rec.lab()].x);
        const float *lc = reinterpret_cast<const float *>(last + rec.coords());
        float p = ((p0.x + p1.x) + p2.x) + a3.x;
        for (int pnt = 0; pnt < (int)(N_total & 3); ++pnt)  // the N mod 4 terms after the lanes' ranges go onto lane 0
            if (((lw >> (8 * pnt)) & 255u) == (unsigned)j) p = p + lc[pnt * kD + i];
        p = p + (((p0.y + p1.y) + p2.y) + a3.y);
        p = p + (((p0.z + p1.z) + p2.z) + a3.z);
        p = p + (((p0.w + p1.w) + p2.w) + a3.w);
        const float c = p / (float)sCntTot[j];  // 0/0 = NaN for an empty cluster (kmeans.py:182)
        const float diff = cen[tid] - c;
        cen[tid] = c;
        sSq[tid] = diff * diff;
    }
    __syncthreads();
    const float error = inner_sum_parallel(sSq, dk, sScr);
    if (tid == 0) {
        double sum = 0.0;
        for (int r = 0; r < P; ++r) {
            const float4 w = table[(int64_t)r * words + rec.sin()];
            sum = sum + __longlong_as_double((long long)(((unsigned long long)__float_as_uint(w.y) << 32) | __float_as_uint(w.x)));
        }
        const float inertia = (float)(-(sum / (double)N_total));
        const int64_t it = state->iter;
        if (a.trace) {
            a.trace[2 * it] = error;
            a.trace[2 * it + 1] = inertia;
        }
        state->inertia = (double)inertia;
        state->error = (double)error;
        state->iter = it + 1;
        state->done = (error <= a.tol) ? 1 : 0;
    }
}

// before the loop: state, working centroids, counters
__global__ __launch_bounds__(kThreads) void reforder_fast_prepare_kernel(const Args a, const float *__restrict__ cen_in) {
    unsigned char *ws = a.ws + (int64_t)blockIdx.x * a.ws_stride;
    const int dk = kD * a.K;
    et_kmeans_state *st = at<et_kmeans_state>(ws, a.lay.state);
    if (threadIdx.x == 0) {  // (max_abs_x / bad_input stay as the scan left them)
        st->n_total = a.geo.N;
        st->iter = 0;
        // non-finite input in ANY problem of the batch stops all of them before the first iteration (the batch iterates and
        // stops jointly: a problem that sat out would leave the others waiting for its arrival); the host reads the
        // bad_input flags after the loop and returns ET_ERR_BAD_DATA
        int bad = 0;
        for (int b = 0; b < a.batch; ++b) bad |= at<et_kmeans_state>(a.ws + (int64_t)b * a.ws_stride, a.lay.state)->bad_input;
        st->done = bad ? 1 : 0;
        st->error = 0.0;
        st->inertia = 0.0;
    }
    for (int e = threadIdx.x; e < dk; e += blockDim.x) at<float>(ws, a.lay.cen)[e] = cen_in[(int64_t)blockIdx.x * dk + e];
    if (threadIdx.x == 0) at<unsigned>(ws, a.lay.arrive)[0] = 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.batch_arrive = 0u;
}

// after the loop: labels in the caller's order (int64), centroids
__global__ __launch_bounds__(kThreads) void reforder_fast_finish_kernel(const Args a, float *__restrict__ cen_out,
                                                                        int64_t *__restrict__ labels) {
    unsigned char *ws = a.ws + (int64_t)blockIdx.y * a.ws_stride;
    const int dk = kD * a.K;
    const Geo &geo = a.geo;
    const int lp = geo.lp;
    const int64_t L = (int64_t)1 << lp, L2 = L * L, N = geo.N;
    if (blockIdx.x == 0)
        for (int e = threadIdx.x; e < dk; e += blockDim.x) cen_out[(int64_t)blockIdx.y * dk + e] = at<float>(ws, a.lay.cen)[e];
    if (!labels) return;
    const uint8_t *LT = at<const uint8_t>(ws, a.lay.LT), *tl = at<const uint8_t>(ws, a.lay.tail);
    int64_t *out = labels + (int64_t)blockIdx.y * N;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        uint8_t v;
        if (n >= geo.tail0) {
            v = tl[n - geo.tail0];
        } else {
            const int64_t g = n / (4 * L2), m = n % (4 * L2);
            const int64_t k = m & 3, lt = m >> 2, c = lt >> lp, r = lt & (L - 1);
            const int64_t q = c >> 4, t = (c & 15) * 4 + k, rb = r >> 2, u = r & 3;
            v = LT[(g * L2 + (q * (L / 4) + rb) * 64 + t) * 4 + u];
        }
        out[n] = (int64_t)v;
    }
}

static bool fast_shape(int64_t N, int d, int K) {
    if (d != kD || K < 1 || K > kFMaxK || N < 1024 || N >= ((int64_t)1 << 29)) return false;
    const Geo g = make_geo(N);
    return g.lp <= kFMaxLp && g.G >= 1;
}
// level-0 tiles (16 chunks) whose accumulators are in LDS at a time: ONE -- at L = 32 two tiles (73 KB, two workgroups per
// CU) took 123 us per iteration at 1e7 points against 111 us with one (38 KB, four per CU), same box
static int fast_tiles_per_round(const Geo &) { return 1; }
static int fast_filter_min_lp() { return options().reforder_filter_min_lp.load(std::memory_order_relaxed); }
static size_t fast_lds_bytes(const Geo &g, int K, int TR) {
    const int L = 1 << g.lp;
    const size_t body = acc_region_bytes(K, L, TR) + sizeof(unsigned) * (size_t)L * L;
    return (body + 15) / 16 * 16;
}
// rows of d K float4 the update kernel stages at a time, and its dynamic LDS
static int update_rows_cap(const Geo &g, int K, int batch, size_t *lds) {
    const size_t row = sizeof(float4) * ((size_t)kD * K + kFMaxK / 4);
    const int L = 1 << g.lp;
    int want = g.full_blk > L ? g.full_blk : L;
    if ((size_t)want * row > kUMaxLds) want = (int)(kUMaxLds / row);
    size_t bytes = (size_t)want * row;
    const size_t sq = sizeof(float) * (size_t)batch * kD * K;
    if (sq > bytes) bytes = sq;
    *lds = (bytes + 15) / 16 * 16;
    return want;
}

#ifdef ET_EXP_RF_CHECK
extern "C" int et_debug_rfcheck(unsigned *host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_rf_check), sizeof(unsigned) * 64) == hipSuccess ? 0 : 3;
}
#endif
#ifdef ET_EXP_RFSTAMP
extern "C" int et_debug_rfstamps(unsigned long long *host) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_rf_stamps), sizeof(unsigned long long) * 64) != hipSuccess) return 3;
    static const unsigned long long zeros[64] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_rf_stamps), zeros, sizeof zeros) == hipSuccess ? 0 : 3;  // (reading resets)
}
#endif

}  // namespace fast

}  // namespace reforder
}  // namespace et
