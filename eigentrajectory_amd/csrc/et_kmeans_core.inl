// et_kmeans_core.inl -- part of csrc/et_kmeans.hip (ONE translation unit: this file is #included there, in order, and is not
// compiled on its own): shared definitions: threads / limits, the scalar helpers the oracle shares, centroids in LDS, the scan / begin kernels and the exact (vector-ALU) assignment scan.
// clang-format off: the fragment starts and ends at namespace scope of whatever the including file has open.
namespace et {

constexpr int kKmThreads = 256;
// The filter kernels run as ONE fat workgroup per CU: few workgroup partials (a single workgroup can fold them and
// update the centroids in one short launch, kmeans_reduce_update_kernel, without any inter-workgroup hand-off).  They
// are compiled for up to 1024 threads and take their size from blockDim.x: the host launches 12 wavefronts (768
// threads, three per SIMD) or 16 (four per SIMD), whichever finishes the shard earlier -- a 256-point pass takes
// 0.73x as long with three wavefronts per SIMD as with four (the kernel is short of instruction-level parallelism,
// not of wavefronts), but a wavefront then has 4/3 as many passes to do, and the count is an integer.  Per Lloyd
// launch at N = 1e7 (same box): 256 threads 60.3 us, 512: 58.0, 640: 60.5, 704: 57.1, 768: 54.7, 832: 59.3, 896: 57.8,
// 960: 58.7, 1024: 57.2 (sizes that load the four SIMDs unevenly lose); at N = 1e6 768 needs two passes, 1024 one.
#ifndef ET_KM_MAXTHREADS
#define ET_KM_MAXTHREADS 1024
#endif
constexpr int kFilterMaxThreads = ET_KM_MAXTHREADS;
constexpr int kFilterMinThreads = 768;
constexpr int kKmMaxBlocks = 4096;

// ---- scalar helpers shared with the oracle's definitions -----------------------------------
__device__ __forceinline__ int bits_for(int64_t n) {  // smallest b with 2^b > n
    return n > 0 ? 64 - __clzll((long long)n) : 0;
}

// trunc(x * 2^frac) for finite x, by shifting the mantissa: bit-identical to the oracle's
// (int64_t)ldexp((double)x, frac) and ~10 integer ops instead of an fp64 -> i64 emulation.
__device__ __forceinline__ long long to_fixed(float x, int frac) {
    // trunc(x 2^frac) as a 64-bit integer (|x 2^frac| < 2^62 by the choice of frac).  Through fp64: (double)x is exact, the
    // scaling by a power of two is exact (the products stay far inside the fp64 range), and the conversion truncates
    // toward zero -- the same integer as shifting the mantissa, without that version's data-dependent branches (six of
    // these per accumulated point: every point in iteration 0, every queued point that changes its label later).
    return (long long)ldexp((double)x, frac);
}

__device__ __forceinline__ bool gt_nanmax(float cand, float best) {  // torch.max: NaN beats everything
    return (cand > best) || (isnan(cand) && !isnan(best));
}

__device__ __forceinline__ unsigned orderable(float f) {  // ascending uint order; NaN -> 0 (torch.argmin)
    if (isnan(f)) return 0u;
    if (f == 0.f) return 0x80000000u;  // -0 and +0 tie, like a float compare
    const unsigned u = (unsigned)__float_as_int(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// threadIdx.x behind an empty asm.  The assignment bodies below run inside the iteration loop of the persistent kernel;
// everything they derive from the thread index alone (lane, wavefront, queue and table addresses, the cluster a lane's
// A operand belongs to, ...) is loop invariant there and would be hoisted in front of the loop and kept live across the
// whole body -- measured: 128 VGPRs + 56 B of scratch memory against 116 and none for the same body as a kernel of its
// own.  A volatile asm is never hoisted or merged, so each inlined call derives these values afresh.
__device__ __forceinline__ unsigned thread_x() {
    unsigned t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// ---- centroids in LDS: row j = {c[0..d-1], |c_j|^2}, pitch = d+1 rounded up to 4 floats ------
__device__ __forceinline__ int cpitch(int d) { return (d + 1 + 3) & ~3; }

__device__ __forceinline__ void stage_centroids(const float *__restrict__ cen, int d, int K, float *sC) {
    const unsigned tx = thread_x();  // (opaque per call: see thread_x)
    const int pitch = cpitch(d);
    for (int j = tx; j < K; j += blockDim.x) {
        float bn = 0.f;
        for (int i = 0; i < d; ++i) {
            const float v = cen[i * K + j];
            sC[j * pitch + i] = v;
            bn = bn + v * v;  // kmeans.py:74 |b|^2: sequential sum of rounded squares
        }
        sC[j * pitch + d] = bn;
    }
}

// similarity of one point to every centroid; returns the arg-max and its value.
template <int D>
__device__ __forceinline__ void best_centroid(const float *x, int d_rt, const float *sC, int K, int &label, float &best) {
    const int d = D ? D : d_rt;
    const int pitch = cpitch(d);
    float an = 0.f;
#pragma unroll
    for (int i = 0; i < (D ? D : ET_KMEANS_MAX_D); ++i)
        if (i < d) an = an + x[i] * x[i];  // kmeans.py:73 |a|^2
    int lb = 0;
    float bv = 0.f;
    for (int j = 0; j < K; ++j) {
        const float *c = sC + j * pitch;
        float y = 0.f;
#pragma unroll
        for (int i = 0; i < (D ? D : ET_KMEANS_MAX_D); ++i)
            if (i < d) y = fmaf(x[i], c[i], y);  // kmeans.py:71
        y = y * 2.0f;                            // :72
        y = y - an;                              // :73
        y = y - c[d];                            // :74
        if (j == 0 || gt_nanmax(y, bv)) {
            bv = y;
            lb = j;
        }
    }
    label = lb;
    best = bv;
}

// best_centroid<6> for the filter kernel's queue drain, where ONE wavefront runs it for a handful of points with
// nothing else to hide latencies behind: four centroids per step, their rows requested from LDS together and their
// fmaf chains interleaved (the plain loop is one LDS round trip + nine dependent operations per centroid: 2.7 us for
// K = 20 against 0.9 us).  The same operations per centroid and the comparisons in centroid order => the same result.
// (j0 .. K: the centroids of a range, in order -- the half-wave form of packed_drain splits the K centroids between two
// lanes and merges their results with the same comparison, earlier range first)
__device__ __forceinline__ void best_centroid6_drain(const float *x, const float *sC, int K, int &label, float &best, int j0 = 0) {
    float an = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) an = an + x[i] * x[i];  // kmeans.py:73 |a|^2
    int lb = j0;
    float bv = 0.f;
    const float4 *s4 = reinterpret_cast<const float4 *>(sC);  // rows of 8 floats: c[0..5], |c|^2, -
    int j = j0;
    for (; j + 4 <= K; j += 4) {
        float4 lo[4], hi[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            lo[u] = s4[2 * (j + u)];
            hi[u] = s4[2 * (j + u) + 1];
        }
        float y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) y[u] = fmaf(x[0], lo[u].x, 0.f);  // kmeans.py:71
#pragma unroll
        for (int u = 0; u < 4; ++u) y[u] = fmaf(x[1], lo[u].y, y[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) y[u] = fmaf(x[2], lo[u].z, y[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) y[u] = fmaf(x[3], lo[u].w, y[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) y[u] = fmaf(x[4], hi[u].x, y[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) y[u] = fmaf(x[5], hi[u].y, y[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            y[u] = y[u] * 2.0f;      // :72
            y[u] = y[u] - an;        // :73
            y[u] = y[u] - hi[u].z;   // :74
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (j + u == j0 || gt_nanmax(y[u], bv)) {
                bv = y[u];
                lb = j + u;
            }
        }
    }
    for (; j < K; ++j) {
        const float4 c0 = s4[2 * j], c1 = s4[2 * j + 1];
        float y = fmaf(x[0], c0.x, 0.f);
        y = fmaf(x[1], c0.y, y);
        y = fmaf(x[2], c0.z, y);
        y = fmaf(x[3], c0.w, y);
        y = fmaf(x[4], c1.x, y);
        y = fmaf(x[5], c1.y, y);
        y = y * 2.0f;
        y = y - an;
        y = y - c1.z;
        if (j == j0 || gt_nanmax(y, bv)) {
            bv = y;
            lb = j;
        }
    }
    label = lb;
    best = bv;
}

// Fast arg-max for d = 6, four points per lane as two packed pairs (v_pk_fma_f32 / v_pk_add_f32:
// the same IEEE operations, two points per instruction) with the next centroid row prefetched from
// LDS while the current one is evaluated.  Only valid when no similarity can be NaN/Inf
// (finite centroids, magnitudes < 1e18: checked once per iteration on the device, state->fast_ok),
// so the NaN rule of torch.max (kmeans.py:156) reduces to a plain `>`; results are bit-identical
// to best_centroid().
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void best_centroid_x4_d6(const f32x2 (&xa)[6], const f32x2 (&xb)[6], const float *sC, int K,
                                                    int (&lb)[4], float (&bv)[4]) {
    f32x2 ana = {0.f, 0.f}, anb = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        ana = ana + xa[i] * xa[i];  // kmeans.py:73
        anb = anb + xb[i] * xb[i];
    }
    const float4 *s4 = reinterpret_cast<const float4 *>(sC);  // row j = s4[2j], s4[2j+1] = {c0..c3},{c4,c5,|c|^2,-}
    float4 n0 = s4[0], n1 = s4[1];
    lb[0] = lb[1] = lb[2] = lb[3] = 0;
    for (int j = 0; j < K; ++j) {
        const float4 p0 = n0, p1 = n1;
        if (j + 1 < K) {
            n0 = s4[2 * j + 2];
            n1 = s4[2 * j + 3];
        }
        const float cc[6] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y};
        f32x2 ya = {0.f, 0.f}, yb = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const f32x2 c = {cc[i], cc[i]};
            ya = __builtin_elementwise_fma(xa[i], c, ya);  // kmeans.py:71
            yb = __builtin_elementwise_fma(xb[i], c, yb);
        }
        ya = ya * 2.0f;  // :72
        yb = yb * 2.0f;
        ya = ya - ana;   // :73
        yb = yb - anb;
        const f32x2 bn = {p1.z, p1.z};
        ya = ya - bn;    // :74
        yb = yb - bn;
        if (j == 0) {
            bv[0] = ya.x;
            bv[1] = ya.y;
            bv[2] = yb.x;
            bv[3] = yb.y;
        } else {
            const bool t0 = ya.x > bv[0], t1 = ya.y > bv[1], t2 = yb.x > bv[2], t3 = yb.y > bv[3];
            bv[0] = t0 ? ya.x : bv[0];
            lb[0] = t0 ? j : lb[0];
            bv[1] = t1 ? ya.y : bv[1];
            lb[1] = t1 ? j : lb[1];
            bv[2] = t2 ? yb.x : bv[2];
            lb[2] = t2 ? j : lb[2];
            bv[3] = t3 ? yb.y : bv[3];
            lb[3] = t3 ? j : lb[3];
        }
    }
}

// ------------------------------------------------------------------------------------------
// scan: max |x| and a non-finite flag, straight into the state block (zeroed by the host side)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kKmThreads) void kmeans_scan_kernel(const float *__restrict__ X, int64_t count,
                                                                 et_kmeans_state *state) {
    float m = 0.f;
    unsigned mn = 0x7f800000u;  // bits of the smallest non-zero |x| (positive floats order like their bits)
    int bad = 0;
    auto take = [&](float v) {
        const float a = fabsf(v);
        if (!(a <= 3.402823466e+38f)) bad = 1;
        else {
            if (a > m) m = a;
            const unsigned b = (unsigned)__float_as_int(a);
            if (b != 0u && b < mn) mn = b;
        }
    };
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // 16-B loads over the aligned body, scalar loads for the (< 4 element) head and tail
    const int64_t head = min(count, (int64_t)(((16 - (reinterpret_cast<uintptr_t>(X) & 15u)) & 15u) / 4));
    const int64_t n4 = (count - head) / 4;
    const float4 *X4 = reinterpret_cast<const float4 *>(X + head);
    for (int64_t i = tid; i < n4; i += stride) {
        const float4 v = X4[i];
        take(v.x);
        take(v.y);
        take(v.z);
        take(v.w);
    }
    if (tid < head) take(X[tid]);
    if (tid < count - head - 4 * n4) take(X[head + 4 * n4 + tid]);
    for (int o = 32; o > 0; o >>= 1) {
        m = fmaxf(m, __shfl_xor(m, o));
        const unsigned other = (unsigned)__shfl_xor((int)mn, o);
        mn = other < mn ? other : mn;
        bad |= __shfl_xor(bad, o);
    }
    // one set of device-scope atomics per WORKGROUP (they serialise on their three addresses)
    __shared__ float sM[kKmThreads / 64];
    __shared__ unsigned sMn[kKmThreads / 64];
    __shared__ int sBad[kKmThreads / 64];
    if ((threadIdx.x & 63) == 0) {
        sM[threadIdx.x >> 6] = m;
        sMn[threadIdx.x >> 6] = mn;
        sBad[threadIdx.x >> 6] = bad;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kKmThreads / 64; ++w) {
            m = fmaxf(m, sM[w]);
            mn = sMn[w] < mn ? sMn[w] : mn;
            bad |= sBad[w];
        }
        // max / min only ever move one way: a workgroup whose value would not move them (by a possibly stale look at the
        // current one -- the worst case is an unnecessary atomic) leaves them alone; ~1000 same-address device atomics
        // at ~15 ns each were a third of this kernel
        const unsigned long long mbits = (unsigned long long)__double_as_longlong((double)m);
        unsigned long long *pmax = reinterpret_cast<unsigned long long *>(&state->max_abs_x);
        unsigned long long *pmin = reinterpret_cast<unsigned long long *>(&state->min_nz_x_bits);
        if (mbits > __hip_atomic_load(pmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(pmax, mbits);
        if ((unsigned long long)mn < __hip_atomic_load(pmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMin(pmin, (unsigned long long)mn);
        if (bad) atomicMax(reinterpret_cast<unsigned long long *>(&state->bad_input), 1ull);
    }
}

// state->fast_ok (set by kmeans_begin_kernel and by every update):
// 0: similarities may be NaN/Inf -> NaN-aware scalar path.
// 1: no similarity of the coming assignment can overflow or be NaN (every centroid finite, all
//    magnitudes below 1e18: |2 a.b| + |a|^2 + |b|^2 <= 4 d 1e36 < FLT_MAX for d <= 32).
// 2: additionally every non-zero |x| and |c| is >= 2^-50.  Then every partial sum of the a.b chain is
//    a multiple of 2^-146, i.e. exactly representable even when subnormal, so scaling the chain by two
//    commutes with every rounding: fl(2c.x) == 2 fl(c.x) bit for bit.  The matrix-core kernel relies
//    on that to fold the reference's "y *= 2" (kmeans.py:72) into its A operand.
__device__ __forceinline__ int sim_frac_bits(double mx, double mc, int d, int64_t n_total) {
    const double m = mx > mc ? mx : mc;
    return 62 - exponent_above(4.0 * d * m * m) - bits_for(n_total);
}

// One wavefront: the d K centroid values are looked at by the 64 lanes in parallel (maxima / minima / a flag: order
// independent; a single lane used to walk through them with two dependent global loads per value -- 14 us of a
// kernel that does almost nothing).
// blockIdx.x = problem of a batch (et_kmeans_fit_batch): state blocks ws_stride bytes apart, centroids cen_stride floats
// apart; shared_scan: the problems share their points, problem 0's state holds the scan results for all of them.
__global__ void kmeans_begin_kernel(et_kmeans_state *state, int64_t n_total, const float *__restrict__ cen, int d,
                                    int K, int64_t ws_stride = 0, int64_t cen_stride = 0, int shared_scan = 0) {
    if (threadIdx.x >= 64) return;
    const et_kmeans_state *scanned = state;
    state = reinterpret_cast<et_kmeans_state *>(reinterpret_cast<char *>(state) + (int64_t)blockIdx.x * ws_stride);
    cen += (int64_t)blockIdx.x * cen_stride;
    if (!shared_scan) scanned = state;
    const int lane = threadIdx.x, n = d * K;
    double mc = 0.0;
    unsigned mn = 0x7f800000u;
    int bad = 0;
    for (int i = lane; i < n; i += 64) {
        const float a = fabsf(cen[i]);
        if (!(a <= 3.402823466e+38f)) bad = 1;
        const double ad = (double)a;
        if (ad > mc) mc = ad;  // NaN ignored
        const unsigned b = (unsigned)__float_as_int(a);
        if (a <= 3.402823466e+38f && b != 0u && b < mn) mn = b;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const double om = __shfl_xor(mc, o);
        mc = om > mc ? om : mc;
        const unsigned on = (unsigned)__shfl_xor((int)mn, o);
        mn = on < mn ? on : mn;
        bad |= __shfl_xor(bad, o);
    }
    if (lane != 0) return;
    const double mx = scanned->max_abs_x;
    const int64_t bad_input = scanned->bad_input, min_nz = scanned->min_nz_x_bits;
    state->max_abs_x = mx;  // (the same values when the state is its own scan result)
    state->bad_input = bad_input;
    state->min_nz_x_bits = min_nz;
    state->n_total = n_total;
    state->frac = 62 - exponent_above(mx) - bits_for(n_total);
    state->max_abs_c = mc;
    state->sim_frac = sim_frac_bits(mx, mc, d, n_total);
    int64_t fast = 0;
    if (!bad && mx < 1e18 && mc < 1e18) {
        const unsigned lim = 0x26800000u;  // 2^-50, see the fast_ok levels above
        fast = (mn >= lim && (unsigned long long)min_nz >= lim) ? 2 : 1;
    }
    state->fast_ok = fast;
    state->iter = 0;
    state->done = bad_input ? 1 : 0;  // non-finite data: every later step is a no-op
    state->error = 0.0;
    state->inertia = 0.0;
}

// ------------------------------------------------------------------------------------------
// Lloyd half-step: labels + exact partial sums.  VEC = points per lane per pass (4 when the
// coordinate rows are 16-B aligned, else 1).  Workgroup accumulators live in LDS (64-bit
// integer atomics, order-free); each workgroup writes one partial block, summed afterwards.
// ------------------------------------------------------------------------------------------
// A workgroup's exact partial sums leave the kernel either as one column of the [entry][workgroup] table (folded by
// kmeans_reduce_partials_kernel; what the sharded step API uses) or, in the single-GPU fit, as device-scope integer
// atomics onto kAccLanes copies of the totals (lane = workgroup index mod kAccLanes): at most grid / kAccLanes
// arrivals per address, nothing to fold afterwards except kAccLanes values per entry, and no arrivals at all for the
// entries a workgroup did not change.
constexpr int kAccLanes = 16;
// entries between two compact copies of the delta table (a table has room for kAccLanes * plen entries: the host side
// lowers the number of copies until they fit)
__host__ __device__ __forceinline__ int compact_pitch(int plen) { return (plen + 31) & ~31; }

__device__ __forceinline__ void emit_partials(const long long *sAcc, int plen, int n_threads,
                                              long long *__restrict__ block_partials, long long *__restrict__ lanes,
                                              int copy_mask = kAccLanes - 1) {
    const unsigned tx = thread_x();  // (opaque per call: see thread_x)
    if (lanes) {
        // copy_mask = 15: sixteen copies per entry, [entry][copy]; 0: one copy at the same stride (small persistent
        // grids); -1: one copy, entries adjacent (the sharded loop's wire format)
        // ... -C (C = 2, 4, 8): C compact copies, kCompactPitch entries apart, workgroup b adds onto copy b % C (the chained
        // loop on one GPU: 256 workgroups' arrivals on one address are served one after the other, ~15 ns each)
        const int stride = copy_mask < 0 ? 1 : kAccLanes, mask = copy_mask < 0 ? 0 : copy_mask;
        const int base = copy_mask < -1 ? (int)(blockIdx.x & (unsigned)(-copy_mask - 1)) * compact_pitch(plen) : 0;
        for (int i = tx; i < plen; i += n_threads) {
            const long long v = sAcc[i];
            if (v != 0)
                atomicAdd(reinterpret_cast<unsigned long long *>(&lanes[base + i * stride + (blockIdx.x & mask)]),
                          (unsigned long long)v);
        }
    } else {
        // transposed [entry][workgroup] so that the reduction reads unit-stride
        for (int i = tx; i < plen; i += n_threads) block_partials[(size_t)i * gridDim.x + blockIdx.x] = sAcc[i];
    }
}

struct PackedHeader {  // written by kmeans_pack_kernel
    float mu[6];
    float s;        // power of two
    float mu_norm;  // >= s ||mu||
    int ok;         // 0: scale out of range / non-finite sample: the fp32 filter decides
    int pad[7];
};
struct LloydPacked {
    const unsigned *xh;
    const unsigned short *rr;
    const float4 *xa;
    const PackedHeader *hdr;
    int fused;  // the exact first iteration of the fit writes the copy (default); 0: kmeans_pack_kernel did, before the loop
};
constexpr int kPackSamples = 1024;

// where the exact first iteration of a fit (assign_body_valu<6, 4>) writes the packed copy of the points it reads anyway
struct PackOut {
    unsigned *xh;
    unsigned short *rr;
    float4 *xa;
    float mu[6];
    float s;
};

// mu (the mean of kPackSamples evenly spaced points, the same in every workgroup: fixed order) and the scale of the packed
// copy; every thread of the workgroup calls (two barriers), the first kKmThreads do the work.  -> usable?
__device__ __forceinline__ bool packed_header(const float *__restrict__ X, int64_t N, const et_kmeans_state *__restrict__ state,
                                              PackedHeader *__restrict__ hdr, float (&mu)[6], float &s) {
    constexpr int d = 6;
    __shared__ double sSum[kKmThreads / 64][d];
    __shared__ float sMu[8];
    const unsigned tid = thread_x();
    const int lane = (int)(tid & 63), wave = (int)(tid >> 6);
    if (tid < kKmThreads) {
        double acc[d] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int k = 0; k < kPackSamples / kKmThreads; ++k) {
            const int64_t idx = ((int64_t)(tid + kKmThreads * k) * N) / kPackSamples;
#pragma unroll
            for (int i = 0; i < d; ++i) acc[i] += (double)X[(int64_t)i * N + idx];
        }
#pragma unroll
        for (int i = 0; i < d; ++i) {
            for (int o = 32; o > 0; o >>= 1) acc[i] += __shfl_xor(acc[i], o);
            if (lane == 0) sSum[wave][i] = acc[i];
        }
    }
    __syncthreads();
    if (tid < d) {
        double t = 0.0;
        for (int w = 0; w < kKmThreads / 64; ++w) t += sSum[w][tid];
        sMu[tid] = (float)(t / (double)kPackSamples);
    }
    __syncthreads();
    double mu_max = 0.0, mu_sq = 0.0;
#pragma unroll
    for (int i = 0; i < d; ++i) {
        mu[i] = sMu[i];
        mu_max = fmax(mu_max, fabs((double)mu[i]));
        mu_sq += (double)mu[i] * (double)mu[i];
    }
    const double bound = state->max_abs_x + mu_max;  // >= every |x_i - mu_i|
    const int e = exponent_above(bound);
    const bool ok = state->fast_ok && !state->bad_input && bound == bound && bound < 1e30 && e >= -40 && e <= 60;
    s = ldexpf(1.0f, 4 - e);
    if (blockIdx.x == 0 && tid == 0) {
#pragma unroll
        for (int i = 0; i < d; ++i) hdr->mu[i] = mu[i];
        hdr->s = s;
        hdr->mu_norm = (float)(sqrt(mu_sq) * (double)s * 1.001) + 1e-30f;
        hdr->ok = ok ? 1 : 0;
    }
    return ok;
}

// Where point n's exact coordinates (a row of 32 B) lie in the side-by-side copy `xa`: blocks of 256 points = four planes of
// 64 rows, plane q holding the q-th point of every quad of the block.  The copy is WRITTEN by lanes that own consecutive
// quads, so the two store instructions of a wavefront for its q-th points fill one plane = 2 KB contiguous (rows in point
// order made every store instruction 64 pieces of 16 B, 128 B apart: the pack pass ran at 3.2 TB/s); a queued point's gather
// still reads one 32-byte row.  (Eight planes of 16-byte pieces -- every store instruction 1 KB contiguous -- write as
// fast, but the gather's two pieces 1 KB apart cost the steady launches 0.4 us each.)
__device__ __forceinline__ int64_t xa_index(int64_t n) {
    return (n >> 8) * 512 + (int64_t)(n & 3) * 128 + (int64_t)((n & 255) >> 2) * 2;
}

// the packed form of the four points n .. n + 3 (x[v][i]: coordinate i of point n + v)
__device__ __forceinline__ void pack_quad(const float (&x)[4][6], int64_t n, int64_t N, const PackOut &po) {
    constexpr float kUp = 1.001953125f, kTiny = 1.1920928955078125e-7f;
    unsigned hw[3][4];
    unsigned short rh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float xc[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) xc[i] = x[q][i] - po.mu[i];
        float an = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) an = fmaf(xc[i], xc[i], an);
        const float rs = fmaf(__builtin_amdgcn_sqrtf(an) * po.s, kUp, kTiny);
        const auto rp = __builtin_amdgcn_cvt_pkrtz(fmaf(rs, kUp, kTiny), 0.f);  // survives the rounding toward zero
        rh[q] = (unsigned short)(__builtin_bit_cast(unsigned, rp) & 0xffffu);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            unsigned h;
            asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=&v"(h) : "v"(xc[2 * p]), "v"(po.s));
            asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+&v"(h) : "v"(xc[2 * p + 1]), "v"(po.s));
            hw[p][q] = h;
        }
        const int64_t ia = xa_index(n + q);
        po.xa[ia] = make_float4(x[q][0], x[q][1], x[q][2], x[q][3]);
        po.xa[ia + 1] = make_float4(x[q][4], x[q][5], 0.f, 0.f);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
        *reinterpret_cast<uint4 *>(po.xh + (int64_t)p * N + n) = make_uint4(hw[p][0], hw[p][1], hw[p][2], hw[p][3]);
    *reinterpret_cast<uint2 *>(po.rr + n) =
        make_uint2((unsigned)rh[0] | ((unsigned)rh[1] << 16), (unsigned)rh[2] | ((unsigned)rh[3] << 16));
}

template <int D, int VEC>
__device__ __forceinline__ void assign_body_valu(
    const float *__restrict__ X, int64_t N, int d_rt, int K, const et_kmeans_state *__restrict__ state,
    const float *__restrict__ cen, const int64_t *__restrict__ given, uint8_t *__restrict__ labels,
    long long *__restrict__ block_partials, long long *__restrict__ lanes = nullptr, int copy_mask = kAccLanes - 1,
    const PackOut pack = PackOut{nullptr, nullptr, nullptr, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0.f}) {
    const unsigned tx = thread_x();  // (opaque per call: see thread_x)
    const int d = D ? D : d_rt;
    const int plen = d * K + K + 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    long long *sAcc = reinterpret_cast<long long *>(smem_raw);                       // plen
    float *sC = reinterpret_cast<float *>(smem_raw + sizeof(long long) * ((plen + 1) & ~1));  // K * cpitch

    const int frac = (int)state->frac, sfrac = (int)state->sim_frac;
    // After the first iteration only the points whose label CHANGED touch the accumulators
    // (-x from the old cluster, +x to the new one).  Integer sums make this exact: the running
    // totals are bit-identical to a full re-accumulation, and once Lloyd settles the LDS atomics
    // (the expensive part of this kernel) all but disappear.
    const bool incremental = (state->iter > 0) && (given == nullptr);
    const bool fast = state->fast_ok != 0;
    const int n_threads = (int)blockDim.x;  // 256, or the filter launch size (768 / 1024)
    for (int i = tx; i < plen; i += n_threads) sAcc[i] = 0;
    stage_centroids(cen, d, K, sC);
    __syncthreads();

    long long sim_acc = 0, nan_acc = 0;
    const int64_t n_groups = (N + VEC - 1) / VEC;
    const int64_t stride = (int64_t)gridDim.x * n_threads;
    for (int64_t gidx = (int64_t)blockIdx.x * n_threads + tx; gidx < n_groups; gidx += stride) {
        const int64_t n = gidx * VEC;
        float x[VEC][D ? D : ET_KMEANS_MAX_D];
        unsigned old_packed = 0xffffffffu;
        if (VEC == 4) {
#pragma unroll
            for (int i = 0; i < (D ? D : ET_KMEANS_MAX_D); ++i)
                if (i < d) {
                    const float4 v = *reinterpret_cast<const float4 *>(X + (int64_t)i * N + n);
                    x[0][i] = v.x;
                    x[1 % VEC][i] = v.y;
                    x[2 % VEC][i] = v.z;
                    x[3 % VEC][i] = v.w;
                }
            if (incremental) old_packed = *reinterpret_cast<const unsigned *>(labels + n);
            // the first iteration of a fit that will iterate on the packed copy writes it, from the rows it has just read
            if constexpr (D == 6 && VEC == 4) {
                if (pack.xh) pack_quad(x, n, N, pack);
            }
        } else {
#pragma unroll
            for (int i = 0; i < (D ? D : ET_KMEANS_MAX_D); ++i)
                if (i < d) x[0][i] = X[(int64_t)i * N + n];
            if (incremental) old_packed = labels[n];
        }
        unsigned packed = 0;
        int lbs[4];
        float bests[4];
        if (D == 6 && VEC == 4 && fast && !given) {
            f32x2 xa[6], xb[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                xa[i] = f32x2{x[0][i], x[1 % VEC][i]};
                xb[i] = f32x2{x[2 % VEC][i], x[3 % VEC][i]};
            }
            best_centroid_x4_d6(xa, xb, sC, K, lbs, bests);
            // Full accumulation (first iteration) with every point of this wavefront pass in ONE cluster -- the
            // usual picture right after a farthest-first initialisation on heavy-tailed data, and for any input
            // stored cluster by cluster: sum the lane's four points, reduce over the wavefront, 7 LDS atomics
            // instead of 7 x 256 on one address.  Integer sums: the same totals in any order.
            if (!incremental && __ballot(1) == ~0ull) {
                const int L0 = __builtin_amdgcn_readfirstlane(lbs[0]);
                if (__all(lbs[0] == L0 && lbs[1] == L0 && lbs[2] == L0 && lbs[3] == L0)) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        long long f = to_fixed(x[0][i], frac) + to_fixed(x[1 % VEC][i], frac) + to_fixed(x[2 % VEC][i], frac) +
                                      to_fixed(x[3 % VEC][i], frac);
                        for (int o = 32; o > 0; o >>= 1) f += __shfl_xor(f, o);
                        if ((tx & 63) == 0)
                            atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[i * K + L0]), (unsigned long long)f);
                    }
                    if ((tx & 63) == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[d * K + L0]), 256ull);
#pragma unroll
                    for (int v = 0; v < 4; ++v) sim_acc += to_fixed(bests[v], sfrac);
                    *reinterpret_cast<unsigned *>(labels + n) = (unsigned)L0 * 0x01010101u;
                    continue;
                }
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            int lb;
            float best;
            if (given) {
                lb = (int)given[n + v];
                best = 0.f;
            } else if (D == 6 && VEC == 4 && fast) {
                lb = lbs[v];
                best = bests[v];
            } else {
                best_centroid<D>(x[v], d, sC, K, lb, best);
            }
            packed |= (unsigned)lb << (8 * v);
            const int old = incremental ? (int)((old_packed >> (8 * v)) & 0xffu) : -1;
            if (lb != old) {
                atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[d * K + lb]), 1ull);
                if (old >= 0) atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[d * K + old]), ~0ull);  // -1
#pragma unroll
                for (int i = 0; i < (D ? D : ET_KMEANS_MAX_D); ++i)
                    if (i < d) {
                        const unsigned long long f = (unsigned long long)to_fixed(x[v][i], frac);
                        atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[i * K + lb]), f);
                        if (old >= 0) atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[i * K + old]), 0ull - f);
                    }
            }
            if (!fast && (isnan(best) || isinf(best))) nan_acc += 1;
            else sim_acc += to_fixed(best, sfrac);
        }
        if (packed != old_packed || !incremental) {
            if (VEC == 4) *reinterpret_cast<unsigned *>(labels + n) = packed;
            else labels[n] = (uint8_t)packed;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        sim_acc += __shfl_xor(sim_acc, o);
        nan_acc += __shfl_xor(nan_acc, o);
    }
    if ((tx & 63) == 0) {
        atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[d * K + K]), (unsigned long long)sim_acc);
        atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[d * K + K + 1]), (unsigned long long)nan_acc);
    }
    __syncthreads();
    emit_partials(sAcc, plen, n_threads, block_partials, lanes, copy_mask);
}

template <int D, int VEC>
__global__ __launch_bounds__(kKmThreads) void kmeans_assign_kernel(
    const float *__restrict__ X, int64_t N, int d_rt, int K, const et_kmeans_state *__restrict__ state,
    const float *__restrict__ cen, const int64_t *__restrict__ given, uint8_t *__restrict__ labels,
    long long *__restrict__ block_partials) {
    if (state->done) return;
    assign_body_valu<D, VEC>(X, N, d_rt, K, state, cen, given, labels, block_partials);
}
