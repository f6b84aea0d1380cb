// et_reforder_fast_update.inl -- part of csrc/et_kmeans_reforder.hip (ONE translation unit: this file is #included there, in order, and is
// not compiled on its own): the fast form: second half of an iteration (levels 2 / 3, the lane combination, new centroids, joint stop of a batch).
// ---- second half: level 2 (workgroup b: block b, its groups' results in group order); the workgroup that arrives last:
//      level 3, the leftovers, the lane combination, the new centroids (kmeans.py:180-182); the last one of the batch: the
//      error over the whole (l, d, K) tensor in ATen's order (kmeans.py:45-51, 232), the stop flag, the next launch's
//      counters.  Rows travel memory -> LDS with every load of a pass in flight at once. ----
// SINGLE (TT = 1024 threads, one workgroup per problem): shards of at most `1024 / slot` blocks (N <= 131 072 at K <= 20) --
// thread group b folds block b straight from memory into LDS and the same workgroup goes on with level 3: no arrival, no
// rows through memory, one small workgroup instead of a grid (-1.5 us per iteration where an iteration is 20 us).
template <int TT, bool SINGLE>
__global__ __launch_bounds__(TT) void reforder_update_kernel2(const Args a, int rows_cap, int slot) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, dk = kD * K;
    unsigned char *ws = a.ws + (int64_t)blockIdx.y * a.ws_stride;
    et_kmeans_state *state = at<et_kmeans_state>(ws, a.lay.state);
    if (state->done) {
        if (a.mail && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)  // (the host stops launching when it reads this)
            __hip_atomic_store(a.mail, (1ull << 63) | (unsigned long long)state->iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    const float *X = a.X + (int64_t)blockIdx.y * a.x_stride;
    const Geo &geo = a.geo;
    const int lp = geo.lp, L = 1 << lp;
    const int64_t N = geo.N;
    float *cen = at<float>(ws, a.lay.cen);
    unsigned *arrive = at<unsigned>(ws, a.lay.arrive);
    const float4 *S1 = at<const float4>(ws, a.lay.S1);
    float4 *S2 = at<float4>(ws, a.lay.S2);
    const float4 *T = at<const float4>(ws, a.lay.T);
    const double *Sin = at<const double>(ws, a.lay.Sin);
    __shared__ double sWsum[16];
    __shared__ int sFlag[2];
    __shared__ float sScr[40];
    float4 *sRows = reinterpret_cast<float4 *>(smem);
    [[maybe_unused]] const int who = blockIdx.x == 0 ? 2 : 9;
    RF_STAMP(who, 0);

    // ---- level 2 ----
    const int blk = SINGLE ? tid / slot : (int)blockIdx.x;
    [[maybe_unused]] const int ltid = SINGLE ? tid % slot : tid;  // column of the row this thread folds
    const int64_t g0 = (int64_t)blk << lp;
    const int ng = (int)((geo.G - g0) < L ? (geo.G - g0) : L);
    float4 a2 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int rowlen = dk + kFMaxK / 4;  // float4 per row of S2: the sums, then the block's points per cluster (bit patterns)
    uint4 c2 = make_uint4(0u, 0u, 0u, 0u);
    const uint4 *cnt4 = at<const uint4>(ws, a.lay.cnt);  // rows of kFMaxK counts = kFMaxK / 4 words of 16 bytes
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rS2 = rsrc_of(S2, (int64_t)sizeof(float4) * (geo.n_blk + 1) * rowlen);
    if constexpr (SINGLE) {
        // thread group `blk` (slot threads, ltid = column): its block's rows straight from memory, sixteen in flight, added in
        // row order; the result is row `blk` of the LDS table level 3 reads below
        if (blk < geo.n_blk && ltid < rowlen) {
            const float4 *src = S1 + g0 * dk;
            const uint4 *csrc = cnt4 + g0 * (kFMaxK / 4);
            for (int r8 = 0; r8 < ng; r8 += 16) {
                float4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int r = r8 + u < ng ? r8 + u : r8;
                    v[u] = ltid < dk ? src[r * dk + ltid] : __builtin_bit_cast(float4, csrc[r * (kFMaxK / 4) + (ltid - dk)]);
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (r8 + u < ng) {
                        if (ltid < dk) {
                            a2.x = a2.x + v[u].x;
                            a2.y = a2.y + v[u].y;
                            a2.z = a2.z + v[u].z;
                            a2.w = a2.w + v[u].w;
                        } else {
                            const uint4 c = __builtin_bit_cast(uint4, v[u]);
                            c2.x += c.x;
                            c2.y += c.y;
                            c2.z += c.z;
                            c2.w += c.w;
                        }
                    }
                }
            }
            if (ltid >= dk && blk == 0) {  // block 0 takes the tail's counts along
                const uint4 v = cnt4[geo.G * (kFMaxK / 4) + (ltid - dk)];
                c2.x += v.x;
                c2.y += v.y;
                c2.z += v.z;
                c2.w += v.w;
            }
            sRows[blk * rowlen + ltid] = ltid < dk ? a2 : __builtin_bit_cast(float4, c2);
        }
        __syncthreads();
    } else {
        for (int r0 = 0; r0 < ng; r0 += rows_cap) {
            const int nr = ng - r0 < rows_cap ? ng - r0 : rows_cap;
            const float4 *src = S1 + (g0 + r0) * dk;
            const uint4 *csrc = cnt4 + (g0 + r0) * (kFMaxK / 4);
            for (int r8 = 0; r8 < nr; r8 += 16) {  // sixteen rows' loads in flight per thread: thread = column, rows in sequence
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
            } else if (tid < rowlen) {  // (integers: any order)
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
        if (tid < dk) st16_sc1(rS2, (unsigned)(((int64_t)blk * rowlen + tid) * sizeof(float4)), a2);
        if (tid >= dk && tid < rowlen) {
            if (blk == 0) {  // block 0 takes the tail's counts along
                const uint4 v = cnt4[geo.G * (kFMaxK / 4) + (tid - dk)];
                c2.x += v.x;
                c2.y += v.y;
                c2.z += v.z;
                c2.w += v.w;
            }
            st16_sc1(rS2, (unsigned)(((int64_t)blk * rowlen + tid) * sizeof(float4)), __builtin_bit_cast(float4, c2));
        }
        RF_STAMP(who, 1);
        // ---- arrival: the stores have been performed at the memory side ----
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) sFlag[0] = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)geo.n_blk - 1u;
        __syncthreads();
        RF_STAMP(who, 2);
        if (!sFlag[0]) return;
    }

    // ---- last workgroup of this problem: level 3 over the complete blocks, in block order ----
    RF_STAMP(3, 0);
    float4 a3 = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned ctot[4] = {0u, 0u, 0u, 0u};
    __shared__ unsigned sCntTot[kFMaxK];
    for (int r0 = 0; r0 < geo.full_blk; r0 += rows_cap) {
        const int nr = geo.full_blk - r0 < rows_cap ? geo.full_blk - r0 : rows_cap;
        const unsigned base = (unsigned)((int64_t)r0 * rowlen * sizeof(float4));
        if constexpr (!SINGLE) {
            for (int e0 = 0; e0 < nr * rowlen; e0 += 16 * TT) {  // sixteen 16-byte loads per lane in flight
                float4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int e = e0 + u * TT + tid;
                    v[u] = ld16_sc1(rS2, base + (unsigned)((e < nr * rowlen ? e : 0) * sizeof(float4)));
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int e = e0 + u * TT + tid;
                    if (e < nr * rowlen) sRows[e] = v[u];
                }
            }
            __syncthreads();
        }  // (SINGLE: the rows are in the table already, all of them: rows_cap >= n_blk)
        if (tid < dk) {
            for (int b = 0; b < nr; ++b) {
                const float4 v = sRows[b * rowlen + tid];
                a3.x = a3.x + v.x;
                a3.y = a3.y + v.y;
                a3.z = a3.z + v.z;
                a3.w = a3.w + v.w;
            }
        } else if (tid < rowlen) {
            for (int b = 0; b < nr; ++b) {
                const float4 v = sRows[b * rowlen + tid];
                ctot[0] += __float_as_uint(v.x);
                ctot[1] += __float_as_uint(v.y);
                ctot[2] += __float_as_uint(v.z);
                ctot[3] += __float_as_uint(v.w);
            }
        }
        __syncthreads();
    }
    float4 part_row = make_float4(0.f, 0.f, 0.f, 0.f);  // the partial block's row, this thread's column
    if (geo.n_blk > geo.full_blk && tid < rowlen)
        part_row = SINGLE ? sRows[geo.full_blk * rowlen + tid]
                          : ld16_sc1(rS2, (unsigned)(((int64_t)geo.full_blk * rowlen + tid) * sizeof(float4)));
    if (tid >= dk && tid < rowlen) {
        if (geo.n_blk > geo.full_blk) {
            const float4 v = part_row;
            ctot[0] += __float_as_uint(v.x);
            ctot[1] += __float_as_uint(v.y);
            ctot[2] += __float_as_uint(v.z);
            ctot[3] += __float_as_uint(v.w);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) sCntTot[4 * (tid - dk) + u] = ctot[u];
    }
    // the inertia of this assignment (kmeans.py:234; only printed by the reference): fp64, a fixed order
    // (always as kUThreads = 256 threads would do it -- four wavefronts' partial sums --, so that both forms give the same bits)
    double part = 0.0;
    if (tid < kUThreads) {
        for (int64_t gb = 0; gb <= geo.G; gb += 8 * kUThreads) {  // eight loads in flight; a fixed order per thread
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t g = gb + (int64_t)u * kUThreads + tid;
                v[u] = Sin[g <= geo.G ? g : 0];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (gb + (int64_t)u * kUThreads + tid <= geo.G) part = part + v[u];
        }
    }
    part = wave_sum_f64(part);
    if (lane == 0) sWsum[wave] = part;
    float *sq_mine = a.sq_all + (int64_t)blockIdx.y * dk;
    float *sSq = reinterpret_cast<float *>(smem);
    __syncthreads();  // sCntTot, sWsum
    if (tid < dk) {
        const int j = tid % K;
        const float *x = X + (int64_t)(tid / K) * N;
        const float4 p2 = part_row;
        const float4 p1 = T[tid], p0 = T[dk + tid];
        const unsigned lw = __float_as_uint(T[2 * dk].x);
        float p = ((p0.x + p1.x) + p2.x) + a3.x;
        // the N mod 4 terms after the lanes' ranges go onto lane 0 (their labels: one word from the tail's workgroup)
        for (int64_t n = N / 4 * 4; n < N; ++n)
            if (((lw >> (8 * (int)(n & 3))) & 255u) == (unsigned)j) p = p + x[n];
        p = p + (((p0.y + p1.y) + p2.y) + a3.y);
        p = p + (((p0.z + p1.z) + p2.z) + a3.z);
        p = p + (((p0.w + p1.w) + p2.w) + a3.w);
        const float c = p / (float)sCntTot[j];  // 0/0 = NaN for an empty cluster (kmeans.py:182)
        const float diff = cen[tid] - c;
        cen[tid] = c;
        if (a.batch > 1) __hip_atomic_store(&sq_mine[tid], diff * diff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else sSq[tid] = diff * diff;
    }
    __syncthreads();
    if (tid == 0) {
        double s = sWsum[0];
        for (int w = 1; w < kUThreads / 64; ++w) s = s + sWsum[w];
        __hip_atomic_store(&state->inertia, (double)(float)(-(s / (double)N)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    RF_STAMP(3, 1);
    if (a.batch > 1) {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0)
            sFlag[1] = __hip_atomic_fetch_add(a.batch_arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)a.batch - 1u;
        __syncthreads();
        if (!sFlag[1]) return;
        const int tot = a.batch * dk;
        for (int e = tid; e < tot; e += TT) sSq[e] = __hip_atomic_load(&a.sq_all[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    }
    RF_STAMP(3, 2);
    const float error = inner_sum_parallel(sSq, a.batch * dk, sScr);
    const int done = (error <= a.tol) ? 1 : 0;
    RF_STAMP(3, 3);
    for (int b = tid; b < a.batch; b += TT) {
        et_kmeans_state *st = at<et_kmeans_state>(a.ws + (int64_t)b * a.ws_stride, a.lay.state);
        const int64_t it = st->iter;
        const double ine = __hip_atomic_load(&st->inertia, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.trace) {
            float *tr = a.trace + ((int64_t)b * a.max_iter + it) * 2;
            tr[0] = error;
            tr[1] = (float)ine;
        }
        st->error = (double)error;
        st->iter = it + 1;
        st->done = done;
        if (b == 0 && a.mail)
            __hip_atomic_store(a.mail, ((unsigned long long)(done != 0) << 63) | (unsigned long long)(it + 1), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int b = 0; b < a.batch; ++b) {
        unsigned char *wb = a.ws + (int64_t)b * a.ws_stride;
        if (tid == 0) at<unsigned>(wb, a.lay.arrive)[0] = 0u;
    }
    if (tid == 0) *a.batch_arrive = 0u;
    RF_STAMP(3, 4);
}
