// et_kmeans_persist.inl -- part of csrc/et_kmeans.hip (ONE translation unit: this file is #included there, in order, and is not
// compiled on its own): all iterations of a fit in one launch (kmeans_lloyd_persist_kernel), the inertia pass, predict.
// clang-format off: the fragment starts and ends at namespace scope of whatever the including file has open.
// ------------------------------------------------------------------------------------------
// Single-GPU fit, shards the filter takes: ALL Lloyd iterations in ONE launch (persistent workgroups).
//
// What a kernel boundary costs between two iterations, measured on this chip (tools/archive/exp_overlap*.hip,
// profiles/r03b_overlap2.txt): ~7 us for a 256 x 768-thread grid -- ~1.5 us until the next launch's first workgroup
// runs and ~5.5 us until its LAST one does (the dispatcher places ~3000 wavefronts one after the other), on the
// critical path of every iteration.  Here the grid (one workgroup per CU, all co-resident) stays; iterations are
// separated by a grid barrier that needs NO cache fence: everything that crosses workgroups -- the 16-copy delta
// table -- is written with device-scope atomics and read with device-scope (sc1) loads, both served by the memory
// side, so `buffer_wbl2` / `buffer_inv` (measured: 23 us per iteration for the pair at agent scope, the reason a first
// cooperative version in round 2 lost) never appear; a workgroup's arrival is one relaxed atomic after an
// `s_waitcnt vmcnt(0)` + workgroup barrier, the wait one lane polling that counter (~1.1 us from the last arrival to
// everybody running).  Every workgroup then folds the table and applies the update ITSELF, as in the chained kernel
// (identical integers in => identical centroids / error / convergence flag everywhere), but keeps state, centroids and
// running totals in its own LDS across iterations -- nothing but the table travels through memory, and all workgroups
// leave the loop in the same iteration.  Labels are only ever re-read by the workgroup that wrote them (the
// chunk -> workgroup map is fixed and is the same in the exact first pass and in the filter passes).
// Every spin carries a time-out: a workgroup that waits longer than kSpinTimeoutTicks sets *abort and everybody
// leaves; the host then repeats the fit with the chained kernel (co-residency cannot be promised when another process
// shares the GPU; inside this process et_kmeans_fit hands out the CUs, see PersistSlots).
// ------------------------------------------------------------------------------------------
template <typename T>
__host__ __device__ __forceinline__ T *byte_shift(T *p, int64_t bytes) {
    return reinterpret_cast<T *>(reinterpret_cast<char *>(const_cast<std::remove_const_t<T> *>(p)) + bytes);
}

struct LloydPersist {
    const et_kmeans_state *st_in;  // state block after scan / begin
    const float *cen_in;           // initial centroids (d, K)
    et_kmeans_state *st_out;       // final state
    float *cen_out;                // final centroids
    long long *tot_out;            // final totals
    long long *lanes0, *lanes1, *lanes2;  // three 16-copy delta tables, zeroed before the launch
    unsigned *arrive;              // grid barrier: arrivals so far (zeroed before the launch)
    unsigned *abort;               // set by a workgroup whose wait timed out (zeroed before the launch)
    float *last;                   // centroids + sim_frac of the last assignment (for kmeans_inertia_kernel)
    // blockIdx.y = one of several problems run side by side in one launch (et_kmeans_fit_batch: the n_init fits of the
    // sklearn recipe), each with its own workspace of identical layout: byte distance between two problems' workspaces
    // (every pointer above except cen_in lives there, and so does `labels`), element distances of their points (0: the
    // same points) and of their initial centroids
    int64_t ws_stride, x_stride, cen_stride;
};
constexpr unsigned long long kSpinTimeoutTicks = 50000000ull;  // 0.5 s of the 100 MHz s_memrealtime clock


template <int NREGS, bool SIM>
__global__ __launch_bounds__(kFilterMaxThreads) void kmeans_lloyd_persist_kernel(
    const float *__restrict__ X, int64_t N, int K, LloydPersist pa, uint8_t *__restrict__ labels, float tol,
    float *trace, int max_iter) {
    constexpr int d = 6;
    if (blockIdx.y) {  // this problem's points, initial centroids and workspace
        const int64_t y = blockIdx.y;
        const int64_t off = y * pa.ws_stride;
        X += y * pa.x_stride;
        pa.cen_in += y * pa.cen_stride;
        pa.st_in = byte_shift(pa.st_in, off);
        pa.st_out = byte_shift(pa.st_out, off);
        pa.cen_out = byte_shift(pa.cen_out, off);
        pa.tot_out = byte_shift(pa.tot_out, off);
        pa.lanes0 = byte_shift(pa.lanes0, off);
        pa.lanes1 = byte_shift(pa.lanes1, off);
        pa.lanes2 = byte_shift(pa.lanes2, off);
        pa.arrive = byte_shift(pa.arrive, off);
        pa.abort = byte_shift(pa.abort, off);
        pa.last = byte_shift(pa.last, off);
        labels = byte_shift(labels, off);
    }
    constexpr int kMaxK = 32;  // the filter's limit (km_use_filter)
    constexpr int kMaxPlen = d * kMaxK + kMaxK + 2;
    const int plen = d * K + K + 2;
    __shared__ et_kmeans_state sSt;
    __shared__ long long sTot[(kMaxPlen + 1) & ~1];  // running totals of this fit (every workgroup holds the same)
    __shared__ float sCen[d * kMaxK];
    __shared__ int sAbort;
    const bool wg0 = blockIdx.x == 0;
    constexpr int kStateWords = (int)(sizeof(et_kmeans_state) / sizeof(unsigned));
    if ((int)threadIdx.x < kStateWords)
        reinterpret_cast<unsigned *>(&sSt)[threadIdx.x] = reinterpret_cast<const unsigned *>(pa.st_in)[threadIdx.x];
    for (int e = threadIdx.x; e < d * K; e += (int)blockDim.x) sCen[e] = pa.cen_in[e];
    for (int e = threadIdx.x; e < plen; e += (int)blockDim.x) sTot[e] = 0;
    if (threadIdx.x == 0) sAbort = 0;
    __syncthreads();
    // copies of the delta table the workgroups spread their atomics over: 16 (as in the chained kernel) for a full grid,
    // ONE for a small one (<= 64 arrivals per address are absorbed by the memory side while the workgroups finish, and
    // the fold becomes a single load per entry)
    const int copy_mask = gridDim.x <= 64 ? 0 : kAccLanes - 1;
    int it = 0;
    for (;; ++it) {
        ET_STAMP(0);
        if (it > 0) {
            // ---- grid barrier: every workgroup has added the deltas of assignment it - 1 ----
            if (threadIdx.x == 0) {
                const unsigned want = (unsigned)it * gridDim.x;
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                while (__hip_atomic_load(pa.arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    if (__hip_atomic_load(pa.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
                        __builtin_amdgcn_s_memrealtime() - t0 > kSpinTimeoutTicks) {
                        __hip_atomic_store(pa.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        sAbort = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (sAbort) return;
            ET_STAMP(1);
            // ---- fold the 16 copies of the table this assignment filled (sc1 loads: the atomics were performed at the
            //      memory side) onto the running totals, in place: one lane per entry reads and writes it ----
            const long long *lanes = it % 3 == 0 ? pa.lanes0 : (it % 3 == 1 ? pa.lanes1 : pa.lanes2);
            const int total = plen * kAccLanes, n_threads = (int)blockDim.x;
            const bool have_prev = it > 1;
            if (copy_mask == 0) {  // small grid: one copy, one load per entry, one memory round trip
                for (int e = threadIdx.x; e < plen; e += n_threads) {
                    const long long v = (long long)__hip_atomic_load(
                        reinterpret_cast<const unsigned long long *>(lanes) + e * kAccLanes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sTot[e] = ((have_prev && e < plen - 2) ? sTot[e] : 0) + v;
                }
            } else {
                for (int base = 0; base < total; base += n_threads) {
                    const int idx = base + (int)threadIdx.x;
                    long long v = 0;
                    if (idx < total)
                        v = (long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(lanes) + idx,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int o = kAccLanes / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
                    if (idx < total && (idx & (kAccLanes - 1)) == 0) {
                        const int e = idx / kAccLanes;
                        sTot[e] = ((have_prev && e < plen - 2) ? sTot[e] : 0) + v;
                    }
                }
            }
            __syncthreads();
            ET_STAMP(2);
            update_body(&sSt, sTot, d, K, tol, sCen, wg0 ? trace : nullptr, nullptr, wg0 ? pa.last : nullptr);
            __syncthreads();
            ET_STAMP(3);
            if (wg0) {  // the table launch it - 1 read becomes the one assignment it + 1 adds onto
                long long *zero = (it + 2) % 3 == 0 ? pa.lanes0 : ((it + 2) % 3 == 1 ? pa.lanes1 : pa.lanes2);
                for (int i = threadIdx.x; i < total; i += n_threads)
                    __hip_atomic_store(reinterpret_cast<unsigned long long *>(zero) + i, 0ull, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (sSt.done || it >= max_iter) break;  // kmeans.py:239 / the iteration cap (uniform over the grid)
        long long *wr = (it + 1) % 3 == 0 ? pa.lanes0 : ((it + 1) % 3 == 1 ? pa.lanes1 : pa.lanes2);
        // The assignment is compiled as if it were a kernel of its own: its inputs pass through an empty asm, so nothing
        // derived from them is loop invariant and hoisted out of the iteration loop (held live across the whole body,
        // the hoisted values cost 12 VGPRs + 72 B of scratch memory: a private segment is paid for at every wavefront
        // launch and the spills sit in the hot loop)
        const float *Xi = X;
        uint8_t *li = labels;
        int64_t Ni = N;
        int Ki = K;
        asm volatile("" : "+s"(Xi), "+s"(li), "+s"(Ni), "+s"(Ki), "+s"(wr));
        ET_STAMP(4);
        filter_assign_body<NREGS, SIM>(Xi, Ni, Ki, &sSt, sCen, li, nullptr, wr, copy_mask);
        ET_STAMP(5);
        // arrival: this workgroup's atomics (and its table clear, workgroup 0) have been performed -- every wavefront
        // waits for its own outstanding memory operations (s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0); a workgroup-scope
        // release fence would omit the vmcnt), then the workgroup barrier, then one lane counts
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(pa.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wg0) {
        if ((int)threadIdx.x < kStateWords)
            reinterpret_cast<unsigned *>(pa.st_out)[threadIdx.x] = reinterpret_cast<const unsigned *>(&sSt)[threadIdx.x];
        for (int e = threadIdx.x; e < d * K; e += (int)blockDim.x) pa.cen_out[e] = sCen[e];
        for (int e = threadIdx.x; e < plen; e += (int)blockDim.x) pa.tot_out[e] = sTot[e];
    }
}

// Inertia of the LAST assignment of a fit that did not track it per iteration (kmeans.py:234 of that iteration):
// the exact similarity of every point to the centroid its label names, centroids = the ones that assignment was
// made with (`last`, saved by update_body), summed as the same fixed-point integers as in the assignment kernels
// -> the same bits.  One pass over the coordinates per fit instead of fp64 work in every iteration.
template <int D>
__global__ __launch_bounds__(kKmThreads) void kmeans_inertia_kernel(const float *__restrict__ X, int64_t N, int d_rt, int K,
                                                                     const float *__restrict__ last,
                                                                     const uint8_t *__restrict__ labels,
                                                                     long long *__restrict__ sim_total,
                                                                     int64_t ws_stride = 0, int64_t x_stride = 0,
                                                                     const long long *__restrict__ skip = nullptr) {
    if (skip && *skip) return;  // (chained loop: the last launch accumulated the similarity sum itself)
    const int d = D ? D : d_rt;
    if (blockIdx.y) {  // problem of a batch: last / labels / sim_total live in workspaces ws_stride bytes apart
        X += (int64_t)blockIdx.y * x_stride;
        last = byte_shift(last, (int64_t)blockIdx.y * ws_stride);
        labels = byte_shift(labels, (int64_t)blockIdx.y * ws_stride);
        sim_total = byte_shift(sim_total, (int64_t)blockIdx.y * ws_stride);
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *sC = reinterpret_cast<float *>(smem_raw);
    __shared__ long long sSum[2];
    stage_centroids(last, d, K, sC);
    if (threadIdx.x < 2) sSum[threadIdx.x] = 0;
    __syncthreads();
    const int sfrac = (int)*reinterpret_cast<const long long *>(last + ((d * K + 1) & ~1));
    const int pitch = cpitch(d);
    long long acc = 0, bad = 0;
    int64_t n_vec = 0;
    if constexpr (D == 6) {
        // four points per lane through 16-byte loads (one point per lane left the 250 MB pass at 3.6 TB/s)
        const bool vec = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15u) == 0) && ((reinterpret_cast<uintptr_t>(labels) & 3u) == 0);
        n_vec = vec ? N : 0;
        for (int64_t g = (int64_t)blockIdx.x * kKmThreads + threadIdx.x; 4 * g < n_vec; g += (int64_t)gridDim.x * kKmThreads) {
            float4 v[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i] = *reinterpret_cast<const float4 *>(X + (int64_t)i * N + 4 * g);
            const unsigned l4 = *reinterpret_cast<const unsigned *>(labels + 4 * g);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 *c4 = reinterpret_cast<const float4 *>(sC + (int)((l4 >> (8 * q)) & 0xffu) * 8);
                const float4 c0 = c4[0], c1 = c4[1];
                float x[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) x[i] = q == 0 ? v[i].x : (q == 1 ? v[i].y : (q == 2 ? v[i].z : v[i].w));
                float an = 0.f;
#pragma unroll
                for (int i = 0; i < 6; ++i) an = an + x[i] * x[i];  // kmeans.py:73
                float y = fmaf(x[0], c0.x, 0.f);                     // :71
                y = fmaf(x[1], c0.y, y);
                y = fmaf(x[2], c0.z, y);
                y = fmaf(x[3], c0.w, y);
                y = fmaf(x[4], c1.x, y);
                y = fmaf(x[5], c1.y, y);
                y = y * 2.0f;
                y = y - an;
                y = y - c1.z;
                if (isnan(y) || isinf(y)) bad += 1;
                else acc += to_fixed(y, sfrac);
            }
        }
    }
    for (int64_t n = n_vec + (int64_t)blockIdx.x * kKmThreads + threadIdx.x; n < N; n += (int64_t)gridDim.x * kKmThreads) {
        const float *c = sC + (int)labels[n] * pitch;
        float an = 0.f, y = 0.f;
#pragma unroll
        for (int i = 0; i < (D ? D : ET_KMEANS_MAX_D); ++i)
            if (i < d) {
                const float x = X[(int64_t)i * N + n];
                an = an + x * x;       // kmeans.py:73
                y = fmaf(x, c[i], y);  // :71
            }
        y = y * 2.0f;
        y = y - an;
        y = y - c[d];
        if (isnan(y) || isinf(y)) bad += 1;
        else acc += to_fixed(y, sfrac);
    }
    for (int o = 32; o > 0; o >>= 1) {
        acc += __shfl_xor(acc, o);
        bad += __shfl_xor(bad, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(&sSum[0]), (unsigned long long)acc);
        atomicAdd(reinterpret_cast<unsigned long long *>(&sSum[1]), (unsigned long long)bad);
    }
    __syncthreads();
    if (threadIdx.x < 2 && sSum[threadIdx.x] != 0)
        atomicAdd(reinterpret_cast<unsigned long long *>(&sim_total[threadIdx.x]), (unsigned long long)sSum[threadIdx.x]);
}

__global__ void kmeans_inertia_finish_kernel(et_kmeans_state *state, const float *__restrict__ last, int d, int K,
                                             const long long *__restrict__ sim_total, int64_t ws_stride = 0,
                                             const long long *__restrict__ skip = nullptr) {
    if (threadIdx.x != 0) return;
    if (skip && *skip) return;  // (the inertia the finalize kernel's update made from the last launch's sum stands)
    if (blockIdx.x) {  // problem of a batch
        state = byte_shift(state, (int64_t)blockIdx.x * ws_stride);
        last = byte_shift(last, (int64_t)blockIdx.x * ws_stride);
        sim_total = byte_shift(sim_total, (int64_t)blockIdx.x * ws_stride);
    }
    const int sfrac = (int)*reinterpret_cast<const long long *>(last + ((d * K + 1) & ~1));
    float inertia;
    if (sim_total[1] > 0) inertia = __int_as_float(0x7fc00000);
    else inertia = (float)(-(((double)sim_total[0] * ldexp(1.0, -sfrac)) / (double)state->n_total));  // kmeans.py:57
    state->inertia = (double)inertia;
}


// Single-GPU fit: the reduction above and the update in ONE launch.  One entry per WAVEFRONT (the filter kernel
// runs one fat workgroup per CU, so an entry has only a few hundred workgroup partials); the workgroup that
// arrives last at the ticket (release fence -> device-scope atomic -> acquire fence, so the other workgroups'
// totals are visible to it) runs the update.  Few workgroups => few arrivals: they serialise at ~12-25 ns each.
__global__ __launch_bounds__(kKmThreads) void kmeans_reduce_update_kernel(const long long *__restrict__ block_partials,
                                                                          int n_blocks, int plen, et_kmeans_state *state,
                                                                          long long *partials, unsigned *ticket, int d,
                                                                          int K, float tol, float *cen, float *trace,
                                                                          float *last) {
    __shared__ int sLast;
    const int lane = threadIdx.x & 63, e = blockIdx.x * (kKmThreads / 64) + (threadIdx.x >> 6);
    // every load is issued before the first result is looked at: one memory round trip instead of three
    const int64_t done = state->done, iter = state->iter;
    const long long prev = (e < plen && lane == 0) ? partials[e] : 0;
    long long s = 0;
    if (e < plen)
        for (int b = lane; b < n_blocks; b += 64) s += block_partials[(size_t)e * n_blocks + b];
    if (done) return;
    if (e < plen) {
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) partials[e] = ((iter > 0 && e < plen - 2) ? prev : 0) + s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned arrived = atomicAdd(ticket, 1u);
        sLast = arrived == gridDim.x - 1;
        if (sLast) {
            *ticket = 0u;  // ready for the next launch
            __threadfence();
        }
    }
    __syncthreads();
    if (!sLast) return;
    update_body(state, partials, d, K, tol, cen, trace, nullptr, last);
}

__global__ __launch_bounds__(kKmThreads) void kmeans_labels_i64_kernel(const uint8_t *__restrict__ lb, int64_t N,
                                                                       int64_t *__restrict__ out, int64_t ws_stride = 0) {
    if (blockIdx.y) {  // problem of a batch: uint8 labels in workspaces ws_stride bytes apart, int64 rows of N
        lb = byte_shift(lb, (int64_t)blockIdx.y * ws_stride);
        out += (int64_t)blockIdx.y * N;
    }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // four labels per lane: one 4-B load, two 16-B stores (both buffers come 16-B aligned from the allocator)
    const bool vec = ((reinterpret_cast<uintptr_t>(lb) & 3u) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15u) == 0);
    const int64_t n4 = vec ? N / 4 : 0;
    for (int64_t i = tid; i < n4; i += stride) {
        const unsigned p = reinterpret_cast<const unsigned *>(lb)[i];
        longlong2 a, b;
        a.x = p & 0xffu;
        a.y = (p >> 8) & 0xffu;
        b.x = (p >> 16) & 0xffu;
        b.y = p >> 24;
        reinterpret_cast<longlong2 *>(out)[2 * i] = a;
        reinterpret_cast<longlong2 *>(out)[2 * i + 1] = b;
    }
    for (int64_t n = 4 * n4 + tid; n < N; n += stride) out[n] = (int64_t)lb[n];
}

// predict (kmeans.py:261-272): labels int64 + optional max similarity
template <int D>
__global__ __launch_bounds__(kKmThreads) void kmeans_predict_kernel(const float *__restrict__ X, int64_t N, int d_rt,
                                                                    const float *__restrict__ cen, int K,
                                                                    int64_t *__restrict__ labels,
                                                                    float *__restrict__ maxsims, int64_t x_stride) {
    const int d = D ? D : d_rt;
    // blockIdx.y = batch element: data x_stride floats apart (d N: contiguous (B, d, N); 0: the same points for every
    // element), (B, d, K) centroids -> (B, N) outputs
    X += (int64_t)blockIdx.y * x_stride;
    cen += (int64_t)blockIdx.y * d * K;
    if (labels) labels += (int64_t)blockIdx.y * N;
    if (maxsims) maxsims += (int64_t)blockIdx.y * N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *sC = reinterpret_cast<float *>(smem_raw);
    stage_centroids(cen, d, K, sC);
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * kKmThreads;
    for (int64_t n = (int64_t)blockIdx.x * kKmThreads + threadIdx.x; n < N; n += stride) {
        float x[D ? D : ET_KMEANS_MAX_D];
#pragma unroll
        for (int i = 0; i < (D ? D : ET_KMEANS_MAX_D); ++i)
            if (i < d) x[i] = X[(int64_t)i * N + n];
        int lb;
        float best;
        best_centroid<D>(x, d, sC, K, lb, best);
        if (labels) labels[n] = lb;
        if (maxsims) maxsims[n] = best;
    }
}

// ------------------------------------------------------------------------------------------
// farthest-first initialisation (kmeans.py:88-112): one pass per new centroid.
// best[n] = max(best[n], sim(x_n, c_{i-1})); candidate = arg-min over n (first index on ties,
