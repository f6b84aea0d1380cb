// et_kmeans.hip -- BatchKMeans (EigenTrajectory/kmeans.py) for gfx950.
//
//   euc_sim            kmeans.py:59-76    y = (2 a.b - |a|^2) - |b|^2, a.b an fmaf chain
//   kmeanspp           kmeans.py:78-112   farthest-first: running max-similarity + global arg-min
//   get_labels         kmeans.py:143-158  arg-max (first max wins, NaN wins)
//   compute_centroids  kmeans.py:160-198  per-cluster mean, empty cluster -> NaN
//   fit / predict      kmeans.py:200-272
//
// Layout: X is (d, N) d-major, so every coordinate row is a unit-stride stream: lanes read
// 16 B (4 consecutive points) per row.  One Lloyd step reads d*4 B per point once and keeps
// everything else (K centroids, K*(d+1)+2 accumulators) in LDS.
//
// Kernels of a single-GPU fit with d = 6, K <= 32 (the anchor clustering):
//   farthest-first   kmeans_init_step_kernel        one launch per centroid; steps >= 2 skip the coordinate read of
//                                                    points that provably keep their running maximum, steps >= 3 whole
//                                                    tiles of 256 points on a 16-byte summary
//   iteration 0      kmeans_assign_kernel body      exact scan + full accumulation (inside the loop kernel's launch); for
//                                                    shards >= 2^17 points it also writes the packed copy (pack_quad)
//   iterations >= 1  kmeans_assign_filter_kernel    f16 MFMA upper bounds + exact certification of the old label,
//                                                    undecided points through an LDS queue; deltas leave as atomics
//                    kmeans_lloyd_chain_kernel      ... one launch per iteration: every workgroup applies the PREVIOUS
//                                                    iteration's update itself (fold of the delta table of exact totals,
//                                                    means, error, convergence) and then assigns; no serial section.
//                                                    Trace-less fits of big shards: packed_assign_body, the same test on
//                                                    an f16 copy of the points (14 B per point instead of 24); its
//                                                    per-launch tables are made by idle wavefronts beside the update's
//                                                    reductions (packed_tables_side; round 4, with the other prologue
//                                                    trims: profiles/r04k_lloyd_launch_stamps.txt)
//                    kmeans_lloyd_persist_kernel    all iterations in one launch (grid barrier without cache fences):
//                                                    shards <= 32768 points and the side-by-side fits of a batch
//   after the loop   kmeans_inertia_kernel          inertia of the last assignment when no trace was requested
// Everything else (other d / K, given labels, the sharded step API) runs the exact scan and a two-kernel fold + update.
//
// Exactness: the reference sums per-cluster coordinates in fp32 in torch's reduction order,
// which a parallel machine cannot reproduce.  Here every coordinate is converted to a 64-bit
// fixed-point integer (truncation, power-of-two scale => exact) and integers are summed, so the
// result is independent of the order: the same bits for any workgroup schedule, grid size or
// number of GPUs, and identical to the CPU oracle (oracle/et_oracle.c).
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "et_common.h"
#include <sched.h>

#include "et_hostring.h"
#include "et_options.h"
#include "et_mfma_filter.h"

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

// ------------------------------------------------------------------------------------------
// Lloyd half-step for iterations >= 1: matrix-core FILTER + exact certification (d = 6, K <= 32).
//
// The exact arg-max above costs ~12 VALU slots per (point, cluster) pair and that, not the 24 B/point
// read, is what the kernel waits for (SQ counters: 941 VALU instructions per 256 points, VALU busy 85 %).
// From the second iteration on almost every point keeps its label, and PROVING that is much cheaper
// than recomputing it:
//
//  1. f16 MFMA (v_mfma_f32_32x32x16_f16, ~16x the fp32 rate) evaluates t'_j ~ G_j = 2 x.c_j - |c_j|^2 for
//     all clusters: x and 2c, scaled by a power of two sg so that every magnitude is below 32, are split
//     into f16 (hi, lo) pairs (round to nearest: 2^-22 relative, 2^-25 absolute on the denormal grid -- the bounds
//     below are derived with the looser 2^-20 / 2^-24 of a round-toward-zero split);
//     the four partial products per coordinate and the split -|c|^2 occupy 26 of the 32 k-slots of two
//     MFMAs, accumulation is fp32 (<= 28 additions, order unknown: 2^-18.2 of the sum of magnitudes).
//     In scaled units, r = sg ||x||, C_j = sg ||c_j||:
//         |t'_j - G_j|            <= E2 = 2^-17.5 (r + C_j)^2 + 2^-21.7 (r + C_j) + 2^-34
//         |(Y_j + |x|^2) - G_j|   <= E1 = 2^-21 (r + C_j)^2           (fp32 chain of kmeans.py:71-74)
//     so with eps_j = 2^-16 (r + C_j)^2 + 2^-20 (r + C_j) + 2^-32 (> 2 (E1 + E2)):  Y_j + |x|^2 <= t'_j + eps_j.
//     eps_j = eps(r) + r (2^-15 C_j) + (2^-16 C_j^2 + 2^-20 C_j): the cluster-dependent part is linear in
//     (r, 1) and occupies two more k-slots, i.e. the MFMA delivers the UPPER BOUNDS u_j = t'_j + eps_j - eps(r)
//     (r and the coefficients rounded up).  A far-away centroid has a large error but an even more
//     negative u_j, so outliers do not loosen the test for ordinary points.
//  2. per point only the SECOND largest u is needed (no index): top-2 with v_max3/v_med3, 1.4 VALU
//     slots per pair.
//  3. the similarity Y_l of the point's OLD label l is evaluated exactly (one fmaf chain, needed for
//     the inertia anyway).  If  w = Y_l + |x|^2  exceeds  second + eps(r)  (+ the rounding of w), then
//     every cluster whose u is not the largest loses to l strictly, and l itself cannot be among them
//     (w <= u_l + eps(r)): l owns the largest u and is the reference's arg-max, strictly, no tie.  The
//     label is unchanged, nothing is accumulated (the sums are incremental), Y_l goes into the inertia.
//  4. every other point (label may change, or too close to call) is pushed on a per-wavefront LDS queue
//     and later gets the full exact scan, 64 queued points at a time, one per lane: label, exact
//     deltas, inertia -- exactly what kmeans_assign_kernel computes.
//
// The filter can only say "unchanged" when that is provably what the reference computes, so labels,
// sums and inertia stay bit-identical; its cost is ~430 VALU instructions per 256 points.
//
// MFMA layout: rows = clusters (A, loop invariant), columns = points (B).  A wavefront takes 256
// points per pass, lane (half, col) owning points 4 col..4 col+3 of its 128-point half.  Both halves
// of a column must feed the SAME point, so the owner's packed f16 dwords are broadcast across the
// halves with v_permlane32_swap (one VALU op yields both "lower half's value" and "upper half's
// value"); two tiles (lower points, upper points) per component q, and one more swap brings each
// half-wave the two partial (max, second) pairs of its own points.
// ------------------------------------------------------------------------------------------
// per-wavefront queue of undecided points: 8 rows (x[0..5], point index, old label) of kFilterSlots entries;
// < 64 entries are carried over and one component q of a pass adds at most 64
constexpr int kFilterSlots = 128;
constexpr int kFilterQueue = 8 * kFilterSlots;  // 32-bit words per wavefront

// full exact scan of `cnt` (<= 64) queued points, one per lane
template <bool SIM>
__device__ __forceinline__ void filter_drain(const unsigned *q, int cnt, int K, const float *sC,
                                             uint8_t *__restrict__ labels, long long *sAcc, int frac, int sfrac, int lane,
                                             long long &sim_acc) {
    constexpr int d = 6;
    if (lane >= cnt) return;
    const int64_t n = (int64_t)q[6 * kFilterSlots + lane];
    float x[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = __uint_as_float(q[i * kFilterSlots + lane]);
    const int old = (int)q[7 * kFilterSlots + lane];
    int lb;
    float best;
    best_centroid6_drain(x, sC, K, lb, best);
    if (lb != old) {
        labels[n] = (uint8_t)lb;
        atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[d * K + lb]), 1ull);
        atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[d * K + old]), ~0ull);
#pragma unroll
        for (int i = 0; i < d; ++i) {
            const unsigned long long f = (unsigned long long)to_fixed(x[i], frac);
            atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[i * K + lb]), f);
            atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[i * K + old]), 0ull - f);
        }
    }
    if (SIM) sim_acc += to_fixed(best, sfrac);
}

#ifdef ET_EXP_WAITSTAMP  // development aid (tools/waitstamp.py): where a pass of packed_assign_body spends its time, in
// shader cycles (s_memtime) summed over all wavefronts and launches: [0] passes, [1] cycles from a pass's start to the
// arrival of its own (prefetched) rows = the exposed load wait, [2] cycles of whole passes, [3] cycles inside queue drains,
// [4] drains, [5] cycles from kernel start to the first pass, [6] wavefronts
__device__ unsigned long long g_waitstamp[8];
// ... and where a LAUNCH of the chained kernel goes (thread 0 of every workgroup, cycles between consecutive stamps, summed
// over workgroups and launches): [0] workgroup-launches, [1] entry -> prologue loads arrived, [2] fold + barrier, [3] update,
// [4] barrier + publish, [5] tables, matrix operand, barrier, [6] the pass loop, [7] final drain + barrier,
// [8] copies -> one + barrier + emit
__device__ unsigned long long g_prostamp[16];
__shared__ unsigned long long s_ps_last, s_ps_acc[16], s_ws_acc[8];
#define KM_PSTAMP(i)                                                        \
    do {                                                                    \
        if (threadIdx.x == 0) {                                             \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();     \
            s_ps_acc[i] = (i) ? t_ - s_ps_last : 1ull;                      \
            if ((i) == 0)                                                   \
                for (int z_ = 0; z_ < 8; ++z_) s_ws_acc[z_] = 0ull;         \
            s_ps_last = t_;                                                 \
        }                                                                   \
    } while (0)
#define KM_PSTAMP_FLUSH()                                                               \
    do {                                                                                \
        __syncthreads();                                                                \
        if (threadIdx.x == 0) {                                                         \
            for (int i_ = 0; i_ < 9; ++i_) atomicAdd(&g_prostamp[i_], s_ps_acc[i_]);     \
            for (int i_ = 0; i_ < 7; ++i_) atomicAdd(&g_waitstamp[i_], s_ws_acc[i_]);    \
        }                                                                               \
    } while (0)
#else
#define KM_PSTAMP(i)
#define KM_PSTAMP_FLUSH()
#endif

#ifdef ET_PERSIST_STAMPS  // development aid (tools/persist_stamps.py): per workgroup and iteration, 10 ns ticks
// kinds 0..5 (persistent kernel): top (own arrival done), go, folded, updated, body start, body end;
// kinds 6..9 (inside the filter body): operands staged, passes done (wavefront 0), queue drained, deltas emitted
constexpr int kStampIters = 104, kStampKinds = 10;
__device__ unsigned long long g_persist_stamps[256 * kStampIters * kStampKinds];
__device__ int g_stamp_it[256];
#define ET_STAMP(kind)                                                                                              \
    do {                                                                                                            \
        if (threadIdx.x == 0 && blockIdx.x < 256 && it < kStampIters) {                                             \
            g_stamp_it[blockIdx.x] = it;                                                                            \
            g_persist_stamps[((size_t)blockIdx.x * kStampIters + it) * kStampKinds + (kind)] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                                                           \
    } while (0)
#define ET_BSTAMP(kind)                                                                                             \
    do {                                                                                                            \
        if (threadIdx.x == 0 && blockIdx.x < 256 && g_stamp_it[blockIdx.x] < kStampIters)                           \
            g_persist_stamps[((size_t)blockIdx.x * kStampIters + g_stamp_it[blockIdx.x]) * kStampKinds + (kind)] =  \
                __builtin_amdgcn_s_memrealtime();                                                                   \
    } while (0)
#else
#define ET_STAMP(kind) do { } while (0)
#define ET_BSTAMP(kind) do { } while (0)
#endif

// issue the loads of pass `gg` (256 points: lane (half, col) owns points 4 col .. 4 col + 3 of its 128-point half)
__device__ __forceinline__ void pass_issue(const float *__restrict__ X, int64_t N, const uint8_t *__restrict__ labels, int64_t gg,
                                           int half, int col, float4 (&vn)[6], unsigned &lpn) {
    const int64_t n = gg * 256 + 128 * half + 4 * col;
    // lanes past the end (last pass only) read points 0..3 instead: finite data, results discarded through `valid`
    // (unconditional loads: no exec-masked branch and no zero fill of 25 registers in every pass)
    const int64_t nl = n < N ? n : 0;  // N % 4 == 0
#ifdef ET_EXP_NOLOAD  // measurement aid (tools/ab_lloyd.sh): the assignment without its memory traffic
    const float f = (float)(nl & 1023) * 0.01f;
#pragma unroll
    for (int i = 0; i < 6; ++i) vn[i] = make_float4(f + i, f - i, f * 0.5f, 1.0f + i);
    lpn = 0x01010101u * (unsigned)(nl & 7);
#else
#ifdef ET_EXP_NT_EVERY  // measurement aid: every ET_EXP_NT_EVERY-th pass bypasses the caches (does the rest then stay in the MALL?)
    if ((gg / 12) % ET_EXP_NT_EVERY == ET_EXP_NT_EVERY - 1) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float4 *p = reinterpret_cast<const float4 *>(X + (int64_t)i * N + nl);
            vn[i] = make_float4(__builtin_nontemporal_load(&p->x), __builtin_nontemporal_load(&p->y),
                                __builtin_nontemporal_load(&p->z), __builtin_nontemporal_load(&p->w));
        }
        lpn = *reinterpret_cast<const unsigned *>(labels + nl);
        return;
    }
#endif
#pragma unroll
    for (int i = 0; i < 6; ++i) vn[i] = *reinterpret_cast<const float4 *>(X + (int64_t)i * N + nl);
    lpn = *reinterpret_cast<const unsigned *>(labels + nl);
#endif
}

// The same for a HANDFUL of queued points (what a wavefront of a small shard holds at the end of its one pass: ~1 % of
// 256 points): one point per LANE leaves 60 lanes idle for a serial walk through the K centroids (2.1 us -- a sixth of a
// small shard's whole Lloyd iteration, profiles/r03f_persist_stamps_7e4.txt).  Here a half-wave takes one point and its
// lane j the similarity to centroid j -- the reference's operations in the reference's order, so the same bits --, the
// arg-max is a butterfly over the 32 lanes with torch.max's rule (first maximum wins; the filter body only runs when no
// similarity can be NaN / Inf, fast_ok), lanes 0..5 of the half-wave add the coordinate deltas.  Two points per step.
constexpr int kSmallDrain = 8;
template <bool SIM>
__device__ __forceinline__ void filter_drain_small(const unsigned *q, int cnt, int K, const float *sC,
                                                   uint8_t *__restrict__ labels, long long *sAcc, int frac, int sfrac, int lane,
                                                   long long &sim_acc) {
    constexpr int d = 6;
    const int j = lane & 31, hw = lane >> 5;
    const float4 *s4 = reinterpret_cast<const float4 *>(sC);  // rows of 8 floats: c[0..5], |c|^2, -
    const int jr = j < K ? j : 0;
    const float4 c0 = s4[2 * jr], c1 = s4[2 * jr + 1];
    for (int p0 = 0; p0 < cnt; p0 += 2) {
        const int p = p0 + hw;
        const bool live = p < cnt;
        const int ps = live ? p : 0;
        float x[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) x[i] = __uint_as_float(q[i * kFilterSlots + ps]);
        float an = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) an = an + x[i] * x[i];  // kmeans.py:73 |a|^2
        float y = fmaf(x[0], c0.x, 0.f);                     // :71
        y = fmaf(x[1], c0.y, y);
        y = fmaf(x[2], c0.z, y);
        y = fmaf(x[3], c0.w, y);
        y = fmaf(x[4], c1.x, y);
        y = fmaf(x[5], c1.y, y);
        y = y * 2.0f;   // :72
        y = y - an;     // :73
        y = y - c1.z;   // :74
        int lb = j;
        if (j >= K) y = -__int_as_float(0x7f800000);  // no such centroid: loses to every finite similarity
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            const float oy = __shfl_xor(y, o);
            const int ol = __shfl_xor(lb, o);
            if (oy > y || (oy == y && ol < lb)) {  // first maximum wins (kmeans.py:156: torch.max)
                y = oy;
                lb = ol;
            }
        }
        if (live) {
            const int old = (int)q[7 * kFilterSlots + p];
            if (lb != old) {
                if (j == 0) {
                    labels[(int64_t)q[6 * kFilterSlots + p]] = (uint8_t)lb;
                    atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[d * K + lb]), 1ull);
                    atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[d * K + old]), ~0ull);
                }
                if (j < d) {
                    const float xi = j == 0 ? x[0] : (j == 1 ? x[1] : (j == 2 ? x[2] : (j == 3 ? x[3] : (j == 4 ? x[4] : x[5]))));
                    const unsigned long long f = (unsigned long long)to_fixed(xi, frac);
                    atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[j * K + lb]), f);
                    atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[j * K + old]), 0ull - f);
                }
            }
            if (SIM && j == 0) sim_acc += to_fixed(y, sfrac);
        }
    }
}

// SIM = false: the similarity sum (the inertia of THIS assignment, kmeans.py:234) is not accumulated -- a fit that
// does not record the per-iteration trace evaluates the inertia once, after its last assignment
// (kmeans_inertia_kernel); the labels and the cluster sums are the same either way.
template <int NREGS, bool SIM = true>
__device__ __forceinline__ void filter_assign_body(const float *__restrict__ X, int64_t N, int K,
                                                   const et_kmeans_state *state, const float *cen,
                                                   uint8_t *__restrict__ labels, long long *__restrict__ block_partials,
                                                   long long *__restrict__ lanes = nullptr,
                                                   int copy_mask = kAccLanes - 1) {
    const unsigned tx = thread_x();  // (opaque per call: see thread_x)
    constexpr int d = 6;
    // power-of-two scale: every |x| sg, |c| sg < 32, so that |2c x| sg^2 < 6 * 2^11 and |c|^2 sg^2 < 6 * 2^10 fit
    // f16 and stay far above the -60000 that pads the rows of clusters >= K
    const int e_max = exponent_above(fmax(state->max_abs_x, state->max_abs_c));
    // first iteration (no labels yet), possible NaN/Inf, or a scale whose square leaves the fp32 range:
    // the exact kernel decides
    if (state->iter <= 0 || !state->fast_ok || e_max < -40 || e_max > 60) {
        assign_body_valu<6, 4>(X, N, d, K, state, cen, nullptr, labels, block_partials, lanes, copy_mask);
        return;
    }
    const int plen = d * K + K + 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    long long *sAcc = reinterpret_cast<long long *>(smem_raw);                                 // plen
    float *sC = reinterpret_cast<float *>(smem_raw + sizeof(long long) * ((plen + 1) & ~1));  // K * 8
    const int lane = tx & 63, wave = tx >> 6, half = lane >> 5, col = lane & 31;
    unsigned *queue = reinterpret_cast<unsigned *>(sC + K * 8) + wave * kFilterQueue;
    const int frac = (int)state->frac, sfrac = (int)state->sim_frac;
    // threshold polynomial in rr, already multiplied by sg^2 (the MFMA works on scaled operands)
    constexpr float kUp = 1.001953125f, kTiny = 1.1920928955078125e-7f;  // (1 + 2^-9) v + 2^-23 survives the rtz to f16
    // Trace-less fits (SIM = false) certify with w' = 2 x.c_l - |c_l|^2 instead of the reference's Y_l + |x|^2 (no |x|^2
    // chain, no square root).  With u = 2^-24, a = 2 d - an, Y_l = fl(fl(a) - cc):  Y_l + an = (2 d - cc) + a d1 +
    // (fl(a) - cc) d2, |d1|, |d2| <= u, so |Y_l + an - (2 d - cc)| <= u (2 |a| + cc)(1 + u) <= 3 u (r + C)^2 (1 + 2^-18)
    // <= 2^-21 (r^2 + C^2) (1 + 2^-10); w' itself adds one rounding, u |w'|.  r is bounded by sqrt(6) max|x_i|.
    constexpr float kSqrt6Up = 2.4543f;       // sqrt(6) (1 + 2^-9)
    constexpr float kR2Slack = 1.57365e-5f;   // 2^-16 + 2^-21 (1 + 2^-10), rounded up: eps(r)'s r^2 term + the slack above
    constexpr float kCcSlack = 4.7731e-7f;    // 2^-21 (1 + 2^-10), rounded up
    const float sg = ldexpf(1.0f, 5 - e_max), sg2 = sg * sg;
    const float sgk = sg * kSqrt6Up;  // exact (sg is a power of two, 2^-55 ... 2^45)
    const int n_thr = (int)blockDim.x, n_wav = n_thr >> 6;  // 768 or 1024 threads (host's choice)
    for (int i = tx; i < plen; i += n_thr) sAcc[i] = 0;
    stage_centroids(cen, d, K, sC);
    // slot 7 of a centroid row: the |c|^2 part of the rounding slack of the trace-less certification (below)
    for (int j = tx; j < K; j += n_thr) sC[j * 8 + 7] = fmaf(sC[j * 8 + 6] * sg2, kCcSlack, 2.3283064365386963e-10f);
    __shared__ int sNext;  // next of this workgroup's passes: the wavefronts take them as they come (see the loop)
    if (tx == 0) sNext = (int)(blockDim.x >> 6);  // a wavefront's first pass is its own (static), the others are handed out
    __syncthreads();


    // A operands: this lane feeds accumulator row m = col, k-half = half.  Row m is read back by lanes
    // of half (m >> 2) & 1 in register 4 (m >> 3) + (m & 3); cluster j sits in register j >> 1 of half j & 1,
    // so both halves reduce over registers 0 .. ceil(K / 2) - 1 <= NREGS - 1 (rows of clusters >= K: -60000).
    u32x4 a1 = {0u, 0u, 0u, 0u}, a2 = {0u, 0u, 0u, 0u};
    {
        const int j = 2 * (4 * (col >> 3) + (col & 3)) + ((col >> 2) & 1);
        unsigned ch[3] = {0u, 0u, 0u}, cl[3] = {0u, 0u, 0u};
        float nb = -60000.0f;
        if (j < K) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
                split_f16(sC[j * 8 + 2 * p], sC[j * 8 + 2 * p + 1], 2.0f * sg, ch[p], cl[p]);
            nb = -sC[j * 8 + 6] * sg2;
        }
        // -|c|^2 = hi + lo; lo is carried as lo * 2^10 against a 2^-10 on the point side (finer f16 grid)
        const auto nh = __builtin_amdgcn_cvt_pkrtz(nb, 0.f);
        const unsigned bnd = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(nb, (nb - (float)nh[0]) * 1024.0f));
        // the cluster-dependent part of the error bound, eps_j - eps(r) = r (2^-15 C_j) + (2^-16 C_j^2 + 2^-20 C_j)
        // with C_j = sg ||c_j|| rounded up, rides along as two more k-slots against (r, 1)
        unsigned ebd = 0u;
        if (j < K) {
            const float cj = sqrtf(sC[j * 8 + 6]) * sg * 1.001f;
            ebd = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(fmaf(3.0517578125e-5f * cj, kUp, kTiny),
                                                                           fmaf(fmaf(cj, 1.52587890625e-5f, 9.5367431640625e-7f) * cj, kUp, kTiny)));
        }
        // k-slots: the lower half-wave's lanes meet the points' hi parts (and the {1, 2^-10} of -|c|^2), the upper
        // half-wave's lanes the lo parts (and (r, 1)); the first MFMA multiplies both by hi(2c), the second by lo(2c)
        a1 = u32x4{ch[0], ch[1], ch[2], half == 0 ? bnd : ebd};
        a2 = u32x4{cl[0], cl[1], cl[2], 0u};
    }
    const f16x8 A1 = __builtin_bit_cast(f16x8, a1), A2 = __builtin_bit_cast(f16x8, a2);
    ET_BSTAMP(6);
    const float4 *s4 = reinterpret_cast<const float4 *>(sC);

    // Inertia: trunc(Y 2^sim_frac) is an integer below 2^(62 - bits(n_total)); a lane may add 2^(bits - 9) of them
    // in fp64 without leaving the exactly representable integers (< 2^53), four fp64 ops per point instead of
    // the ~25 of the integer conversion.  Flushed into the 64-bit accumulator before that limit.
    long long sim_acc = 0;
    double dsum = 0.0;
    const double sim_scale = ldexp(1.0, sfrac);
    const int term_limit = 1 << min(max(bits_for(state->n_total) - 9, 2), 30);
    int terms = 0;
    int qn = 0;  // wave-uniform number of queued points
    const int64_t n_groups = (N + 255) / 256;
    // The workgroup's passes (the same set as with a fixed wavefront -> pass map) are handed out through an LDS counter:
    // a SIMD serves its oldest wavefront first, so with a fixed map the first wavefront of a SIMD finished its share at
    // 27 us of a 41 us assignment phase and the SIMD ran its tail with two, then one wavefront.  Sums are exact integers
    // and labels per point, so who processes a pass does not matter.
    // (shards with at most one pass per wavefront keep the fixed map: nothing to balance, and no counter round trip)
    const bool dynamic = n_groups > (int64_t)gridDim.x * n_wav;
    // (Spreading the passes of the last, partial round evenly over all workgroups -- 152.6 each instead of 156 for
    // workgroups 0..182 and 144 for the rest at N = 1e7 -- was measured and dropped: +0.9 us per launch.  A launch's
    // workgroups start over ~5.7 us in index order, so the ones with the extra round are the ones that start first.
    // Leaning into that -- run lengths of the last rounds falling linearly with the workgroup index, 0.04 ... 0.16 passes
    // per index -- lost as well: 50.7 / 50.6 / 51.5 / 52.2 against 50.2 us.)
    bool first = true;
    auto take = [&]() -> int64_t {  // this wavefront's next pass, or -1 (wave-uniform)
        int64_t g;
        if (first) {
            first = false;
            g = (int64_t)blockIdx.x * n_wav + wave;
        } else if (dynamic) {
            int i = 0;
            if (lane == 0) i = atomicAdd(&sNext, 1);
            i = __builtin_amdgcn_readfirstlane(i);
            g = (int64_t)blockIdx.x * n_wav + (i % n_wav) + (int64_t)(i / n_wav) * gridDim.x * n_wav;
        } else {
            g = n_groups;
        }
        return g < n_groups ? g : -1;
    };
    // (Requesting a pass's coordinates one pass ahead, or a wavefront's first pass before the persistent kernel's grid
    // barrier -- 25 more VGPRs each, free at three wavefronts per SIMD -- was measured and dropped: 51.4 against 50.4 us
    // per chained launch at N = 1e7 and no change of the persistent iteration; with the matrix-core work AND the top-2
    // removed the launch still takes 49.5 us (tools/ab_lloyd.sh, profiles/r03e_ab_lloyd.txt): the passes run at what the
    // memory side delivers for this access pattern, ~6 TB/s, and are neither latency nor issue bound.)
    float4 vn[6];
    unsigned lpn = 0u;
    int64_t g = take();
    if (g >= 0) pass_issue(X, N, labels, g, half, col, vn, lpn);
    while (g >= 0) {
        const int64_t n = g * 256 + 128 * half + 4 * col;
        const bool valid = n < N;
        float4 v[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = vn[i];
        const unsigned old_packed = lpn;
        unsigned undecided = 0u;  // bit q: point q of this lane goes to the queue
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float x[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) x[i] = q == 0 ? v[i].x : (q == 1 ? v[i].y : (q == 2 ? v[i].z : v[i].w));
            float an = 0.f, rs;
            if constexpr (SIM) {
#pragma unroll
                for (int i = 0; i < 6; ++i) an = an + x[i] * x[i];  // kmeans.py:73
                rs = fmaf(__builtin_amdgcn_sqrtf(an) * sg, kUp, kTiny);  // >= sg ||x||
            } else {
                const float m = vmax(max3_abs(x[0], x[1], x[2]), max3_abs(x[3], x[4], x[5]));
                rs = fmaf(m, sgk, kTiny);  // = fl(m sg kSqrt6Up + kTiny) >= sg ||x|| as well: ||x|| <= sqrt(6) max |x_i|
            }
            unsigned w[7];  // {xh01, xh23, xh45, xl01, xl23, xl45, (r, 1)}
#pragma unroll
            for (int p = 0; p < 3; ++p) split_f16(x[2 * p], x[2 * p + 1], sg, w[p], w[3 + p]);
            w[6] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(fmaf(rs, kUp, kTiny), 1.0f));
            // One v_permlane32_swap of (hi, lo) dwords yields the B dword of both tiles: r[0] = {lower lanes: own hi,
            // upper lanes: the lower partner's lo} feeds tile L (points of the lower half-wave), r[1] = {lower lanes:
            // the upper partner's hi, upper lanes: own lo} feeds tile U -- no copies, and the same four dwords serve
            // both MFMAs (the second one multiplies the fourth by zero).
            const unsigned ones = 0x14003c00u;  // {1, 2^-10}: partners of {hi, lo * 2^10} of -|c|^2
            u32x4 bLo, bUp;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const auto r = p < 3 ? __builtin_amdgcn_permlane32_swap(w[p], w[3 + p], false, false)
                                     : __builtin_amdgcn_permlane32_swap(ones, w[6], false, false);
                bLo[p] = r[0];
                bUp[p] = r[1];
            }
            const f16x8 BL = __builtin_bit_cast(f16x8, bLo), BU = __builtin_bit_cast(f16x8, bUp);
            f32x16 accL, accU;
#pragma unroll
            for (int r = 0; r < 16; ++r) accL[r] = accU[r] = 0.f;
            float bL, sL, bU, sU;
#ifdef ET_EXP_NOMFMA  // measurement aid: no matrix-core work and no top-2 (every point is "kept")
            bL = __uint_as_float(bLo[0]) * 1e-30f, sL = __uint_as_float(bLo[1]) * 1e-30f - 1e30f;
            bU = __uint_as_float(bUp[2]) * 1e-30f, sU = __uint_as_float(bUp[3]) * 1e-30f - 1e30f;
#else
            accL = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, BL, accL, 0, 0, 0);
            accU = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, BU, accU, 0, 0, 0);
            accL = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2, BL, accL, 0, 0, 0);
            accU = __builtin_amdgcn_mfma_f32_32x32x16_f16(A2, BU, accU, 0, 0, 0);
            top2<NREGS>(accL, bL, sL);
            top2<NREGS>(accU, bU, sU);
#endif
            // lower half-wave: both partials of its own points (tile L); upper half-wave: those of tile U
            const auto rb = __builtin_amdgcn_permlane32_swap(__float_as_uint(bL), __float_as_uint(bU), false, false);
            const auto rq = __builtin_amdgcn_permlane32_swap(__float_as_uint(sL), __float_as_uint(sU), false, false);
            const float b0 = __uint_as_float(rb[0]), b1 = __uint_as_float(rb[1]);
            const float s0 = __uint_as_float(rq[0]), s1 = __uint_as_float(rq[1]);
            const float second = vmed3(b0, b1, vmax(s0, s1));  // second largest upper bound u_j
            // exact similarity to the old label's (updated) centroid, kmeans.py:71-74
            const int ol = (int)((old_packed >> (8 * q)) & 0xffu);
            const float4 r0 = s4[2 * ol], r1 = s4[2 * ol + 1];
            float y = 0.f;
            y = fmaf(x[0], r0.x, y);
            y = fmaf(x[1], r0.y, y);
            y = fmaf(x[2], r0.z, y);
            y = fmaf(x[3], r0.w, y);
            y = fmaf(x[4], r1.x, y);
            y = fmaf(x[5], r1.y, y);
            float wv, th;
            if constexpr (SIM) {
                y = y * 2.0f;
                y = y - an;
                y = y - r1.z;
                // keep <=> (Y_l + |x|^2) sg^2 exceeds every other cluster's upper bound: w - second > eps(r) + rounding of w
                wv = (y + an) * sg2;
                th = fmaf(fabsf(wv), 2.384185791015625e-7f, fmaf(rs, fmaf(rs, 1.52587890625e-5f, 9.5367431640625e-7f), 2.3283064365386963e-10f));
            } else {
                // the same test on a certified lower bound of Y_l + |x|^2 (see kR2Slack above; r1.w = the |c_l|^2 slack)
                wv = fmaf(y, 2.0f, -r1.z) * sg2;
                th = fmaf(fabsf(wv), 1.1920928955078125e-7f, fmaf(rs, fmaf(rs, kR2Slack, 9.5367431640625e-7f), r1.w));
            }
            const bool keep = wv - second > th;
            if (SIM) {
                const double term = trunc((double)y * sim_scale);
                dsum += (valid && keep) ? term : 0.0;
            }
            undecided |= (valid && !keep) ? (1u << q) : 0u;
        }
        if (__ballot(undecided != 0u)) {  // rare once Lloyd settles: queue the points that need the full scan
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool push = (undecided >> q) & 1u;
                const unsigned long long m = __ballot(push);
                if (push) {  // the coordinates travel with the entry: no second trip to HBM for them
                    unsigned *e = queue + qn + __popcll(m & ((1ull << lane) - 1ull));
#pragma unroll
                    for (int i = 0; i < 6; ++i)
                        e[i * kFilterSlots] = __float_as_uint(q == 0 ? v[i].x : (q == 1 ? v[i].y : (q == 2 ? v[i].z : v[i].w)));
                    e[6 * kFilterSlots] = (unsigned)(n + q);
                    e[7 * kFilterSlots] = (old_packed >> (8 * q)) & 0xffu;
                }
                qn += __popcll(m);
#ifdef ET_FILTER_DEBUG
                if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[d * K + K + 1]), (unsigned long long)__popcll(m));
#endif
                if (qn >= 64) {
                    qn -= 64;
                    filter_drain<SIM>(queue + qn, 64, K, sC, labels, sAcc, frac, sfrac, lane, sim_acc);
                }
            }
        }
        if (SIM) {
            terms += 4;
            if (terms + 4 > term_limit) {
                sim_acc += (long long)dsum;
                dsum = 0.0;
                terms = 0;
            }
        }
        g = take();
        if (g >= 0) pass_issue(X, N, labels, g, half, col, vn, lpn);
    }
    sim_acc += (long long)dsum;
    ET_BSTAMP(7);
    if (qn > kSmallDrain) filter_drain<SIM>(queue, qn, K, sC, labels, sAcc, frac, sfrac, lane, sim_acc);
    else if (qn) filter_drain_small<SIM>(queue, qn, K, sC, labels, sAcc, frac, sfrac, lane, sim_acc);
    ET_BSTAMP(8);
    if (SIM) {
        for (int o = 32; o > 0; o >>= 1) sim_acc += __shfl_xor(sim_acc, o);
        if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[d * K + K]), (unsigned long long)sim_acc);
    }
    __syncthreads();
    emit_partials(sAcc, plen, n_thr, block_partials, lanes, copy_mask);
    ET_BSTAMP(9);
}

// ------------------------------------------------------------------------------------------
// Trace-less Lloyd iterations on a PACKED copy of the points (d = 6, K <= 32, big shards).
//
// The filter above reads 24 B per point and iteration to certify that a label did not change, and an iteration takes as
// long as the memory side needs to stream them.  The certification does not need the exact coordinates:
// kmeans_pack_kernel writes, once per fit,
//   xh   three rows of N dwords: the coordinate pairs (0,1), (2,3), (4,5) of  s (x - mu)  rounded to f16 (nearest);
//        mu = the mean of 1024 evenly spaced points (any vector would do: arg-max_j -|x - c_j|^2 does not depend on the
//        origin), s = the power of two that brings every |s (x - mu)| below 16
//   rr   N f16: an upper bound R of  s ||x - mu||
//   xa   32 B per point: the exact coordinates of a point side by side (where: xa_index), for the few points the test cannot decide
// = 14 B per point and iteration instead of 24, and the 1-3 % of undecided points cost one 64-B sector each instead of
// six.  The rounding of x is now by far the largest error of the matrix-core estimate, so the bounds are re-derived
// (scaled units; p = s (x - mu), q_j = s (c_j - mu) exact, R >= ||p||, Q_j >= ||q_j||; r = s ||x|| <= R + m with
// m >= s ||mu||, C_j = s ||c_j||, M_j = m + C_j;  G_j = 2 p.q_j - |q_j|^2, and G_j - G_l = s^2 (Y_j - Y_l) in exact
// arithmetic whatever mu is):
//   |xh_i - p_i|  <= (2^-11 + 2^-23) |p_i| + 2^-25          (x - mu in fp32, then f16 to nearest / denormal grid)
//   the MFMA's  t'_j = sum_i xh_i (ch + cl)_ji - |q_j|^2  (2 q_j split into f16 hi + lo as before, fp32 accumulation
//   of 16 terms):   |t'_j - G_j| <= E2 = 2^-9.99 R Q_j + 2^-17.5 (R + Q_j)^2 + 2^-21 (R + Q_j) + 2^-34
//   the reference's fp32 chain (kmeans.py:71-74):  |s^2 (Y_j + |x|^2) - (G_j + |p|^2 ... )| -- only differences
//   matter --  is within E1_j = 2^-20.99 (R + M_j)^2 of the exact value (the 2^-21 (r + C_j)^2 of the filter above)
//   so  s^2 Y_j - const <= u_j := t'_j + E2_j + E1_j,  and  u_j - epsR(R)  is linear in (R, 1) per cluster: the slope
//   rides in a k-slot against R, the constant is folded into the -|q_j|^2 slots (rounded up).
//   The old label l:  w' = sum_i xh_i (2 s c~_li) - s^2 |c~_l|^2  as an fp32 chain on the f16 values (v_fma_mix_f32),
//   G_l >= w' - Ew_l,  Ew_l = 2^-9.99 R Q_l + 2^-20 (R + Q_l)^2 + 2^-22 Q_l + 2^-40.
//   keep  <=>  w' - second > epsR(R) + Ew_l + E1_l (+ the rounding of the comparison):  then l owns the largest u (were it
//   not, u_l <= second would give w' <= second + epsR + Ew_l) and every other cluster j has  s^2 Y_j - const <= second +
//   epsR < w' - Ew_l - E1_l <= s^2 Y_l - const:  l is the reference's arg-max, strictly.
// Everything else -- the queue, the exact scan of the queued points (now on coordinates fetched from xa), the
// incremental integer sums -- is the filter's; labels, sums and iteration counts stay bit-identical.  Falls back to the
// fp32 filter for an iteration whose centroids leave the packed range (|s (c - mu)| >= 31: cannot happen for means of
// the points, can for caller-provided initial centroids) or when the scale is out of range.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kKmThreads) void kmeans_pack_kernel(const float *__restrict__ X, int64_t N,
                                                                 const et_kmeans_state *__restrict__ state,
                                                                 PackedHeader *__restrict__ hdr, unsigned *__restrict__ xh,
                                                                 unsigned short *__restrict__ rr, float4 *__restrict__ xa) {
    // (stand-alone form, ET_KMEANS_PACK_FUSED=0: by default the exact first iteration of the fit writes the copy)
    constexpr int d = 6;
    PackOut po;
    po.xh = xh;
    po.rr = rr;
    po.xa = xa;
    if (!packed_header(X, N, state, hdr, po.mu, po.s)) return;
    const int tid = threadIdx.x;
    const int64_t n_quads = N / 4;  // N % 4 == 0 (the caller's vec_ok)
    for (int64_t g = (int64_t)blockIdx.x * kKmThreads + tid; g < n_quads; g += (int64_t)gridDim.x * kKmThreads) {
        const int64_t n = 4 * g;
        float x[4][d];
#pragma unroll
        for (int i = 0; i < d; ++i) {
            const float4 v = *reinterpret_cast<const float4 *>(X + (int64_t)i * N + n);
            x[0][i] = v.x;
            x[1][i] = v.y;
            x[2][i] = v.z;
            x[3][i] = v.w;
        }
        pack_quad(x, n, N, po);
    }
}

// a += f16(w.lo or w.hi) * b: the compiler folds the (exact) conversion into one v_fma_mix_f32.  Compiler-visible on
// purpose: these instructions sit between matrix instructions, and the hazard recogniser does not look inside inline
// assembly (an asm helper's output once landed in a register an earlier v_mfma was still reading).
__device__ __forceinline__ float fma_mix_lo(unsigned w, float b, float a) {
    return fmaf((float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu)), b, a);
}
__device__ __forceinline__ float fma_mix_hi(unsigned w, float b, float a) {
    return fmaf((float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)), b, a);
}

constexpr int kPkQueue = 2 * kFilterSlots;  // per wavefront: point index, old label
constexpr int kPkRow = 12;                  // floats per cluster in the label table

#ifndef ET_PK_ACC_COPIES
#define ET_PK_ACC_COPIES 4
#endif
constexpr int kPkAccCopies = ET_PK_ACC_COPIES;  // copies of the workgroup's accumulators in LDS (packed_drain); a power of two
constexpr int kPkAccPitch = 228;   // int64 per copy: >= d K + K + 2 = 226 for K = 32; 456 words = 8 mod 64: eight different banks

// full exact scan of `cnt` (<= 64) queued points, one per lane, on coordinates fetched from the side-by-side copy
__device__ __forceinline__ void packed_drain(const unsigned *q, int cnt, int K, const float *sC, const float4 *__restrict__ xa,
                                             uint8_t *__restrict__ labels, long long *sAcc, int frac, int lane) {
    constexpr int d = 6;
    // At most 32 queued points (the usual case of a wavefront's LAST drain, which sits on the launch's tail with nothing to
    // hide behind): lanes e and e + 32 both fetch point e and scan about half of the centroids each -- the first kh (a multiple
    // of four) and the rest --, then the lower lane merges: the later range wins only by the scan's own comparison
    // (strictly larger, or NaN against non-NaN), i.e. the result is the one of the scan over all K in order.
    const bool halves = __builtin_amdgcn_readfirstlane(cnt) <= 32 && K >= 8;  // (wave-uniform, and known to be)
    const int e = halves ? (lane & 31) : lane;
    const bool mine = e < cnt;
    const int ec = mine ? e : 0;
    const int64_t n = (int64_t)q[ec];
    const int old = (int)q[kFilterSlots + ec];
    const int64_t ia = xa_index(n);
    const float4 a = xa[ia], b = xa[ia + 1];
    const float x[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
    int lb;
    float best;
    const bool up = halves && lane >= 32;
    const int kh = ((K + 4) / 8) * 4;  // K = 20: twelve and eight
    best_centroid6_drain(x, sC, (halves && !up) ? kh : K, lb, best, up ? kh : 0);
    if (halves) {
        const auto rb = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
        const auto rl = __builtin_amdgcn_permlane32_swap((unsigned)lb, (unsigned)lb, false, false);
        const float ub = __uint_as_float(rb[1]);  // (second result, lower lanes: the upper partner's value)
        const bool take = gt_nanmax(ub, best);
        lb = take ? (int)rl[1] : lb;
    }
    if (mine && !up && lb != old) {
        labels[n] = (uint8_t)lb;
        // kPkAccCopies copies of the accumulators, a lane adds onto copy lane % kPkAccCopies: the points that change in
        // one iteration move between a handful of clusters, so the 64 lanes of a drain hit a few addresses each, and LDS
        // atomics of one instruction on the same address are executed one after the other.  Same-box rocprofv3 averages
        // over the bench's 100 iterations: 1 copy 40.3 / 40.5 us, 2: 39.8 / 40.2, 4: 39.7 / 39.8, 8: 39.7 / 40.1 (the
        // iterations in which 3 % of the points move gain 4 us, the quiet ones pay 0.5 us for clearing and folding)
        long long *acc = sAcc + (lane & (kPkAccCopies - 1)) * kPkAccPitch;
        atomicAdd(reinterpret_cast<unsigned long long *>(&acc[d * K + lb]), 1ull);
        atomicAdd(reinterpret_cast<unsigned long long *>(&acc[d * K + old]), ~0ull);
#pragma unroll
        for (int i = 0; i < d; ++i) {
            const unsigned long long f = (unsigned long long)to_fixed(x[i], frac);
            atomicAdd(reinterpret_cast<unsigned long long *>(&acc[i * K + lb]), f);
            atomicAdd(reinterpret_cast<unsigned long long *>(&acc[i * K + old]), 0ull - f);
        }
    }
}

// Dual form of a pass's loads: BOTH lanes of a column request the rows of the column's point in the lower 128-point block
// (-> the B operand of tile L) and in the upper block (tile U) -- the two half-waves ask for the same addresses, the memory
// side sees the bytes once -- instead of exchanging their own rows with v_permlane32_swap (2 issue slots + 2 copies per
// dword).  A lane's own point is the lower block's for half 0, the upper block's for half 1.  Buffer loads: a pass index
// past the end (g < 0: offset 0xfffffff0) or rows past N are out of range, return zeros and cost no traffic, so the request
// needs no branch around it and the two register sets of the loop (unrolled by two: no copies) are waited for by count.
struct PkRows {
    u32x4 vL[3], vU[3];
    unsigned rL[2], rU[2];
    unsigned lp;
};
struct PkSrc {
    __amdgpu_buffer_rsrc_t row[3], rr, lab;
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pk_rsrc(const void *base, int64_t bytes) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(base);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    const int nb = __builtin_amdgcn_readfirstlane((int)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0, nb, 0x00020000);
}
constexpr int64_t kPkDualMaxN = 1ll << 28;  // 4 N bytes per row and every byte offset stay below 2^31
__device__ __forceinline__ void packed_issue_dual(const PkSrc &src, int64_t gg, unsigned lane_off, unsigned own_off, PkRows &o) {
    // byte offset of the lower block's four points of this column inside a row of dwords
    const unsigned oL = gg >= 0 ? (unsigned)gg * 1024u + lane_off : 0xfffffff0u;
    const unsigned oU = gg >= 0 ? oL + 512u : 0xfffffff0u;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        o.vL[p] = __builtin_amdgcn_raw_buffer_load_b128(src.row[p], oL, 0, 0);
        o.vU[p] = __builtin_amdgcn_raw_buffer_load_b128(src.row[p], oU, 0, 0);
    }
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t a = __builtin_amdgcn_raw_buffer_load_b64(src.rr, oL >> 1, 0, 0);
    const u32x2_t b = __builtin_amdgcn_raw_buffer_load_b64(src.rr, gg >= 0 ? oU >> 1 : 0xfffffff0u, 0, 0);
    o.rL[0] = a.x;
    o.rL[1] = a.y;
    o.rU[0] = b.x;
    o.rU[1] = b.y;
    o.lp = __builtin_amdgcn_raw_buffer_load_b32(src.lab, gg >= 0 ? (oL >> 2) + own_off : 0xfffffff0u, 0, 0);
}

// ---- the per-launch tables of packed_assign_body, as functions of a cluster's centroid (c[0..5], |c|^2 as stage_centroids
// sums it): the label-table row and the matrix operand of lane (col, half).  Made either inside packed_assign_body or --
// chained kernel -- by otherwise idle wavefronts beside the update's reductions (packed_tables_side).
__device__ __forceinline__ void pk_table_row(const float (&c)[6], float bn, const float *hdr, float s, float m_up, float *row) {
    constexpr float kUp = 1.001953125f;
    const float s2 = s * s;
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const float ct = c[i] - hdr[i];
        qq = fmaf(ct, ct, qq);
        row[i] = 2.0f * s * ct;
    }
    // (v_sqrt_f32, 1 ulp: both are upper bounds with a 1e-3 margin)
    const float Q = __builtin_amdgcn_sqrtf(qq) * s * 1.001f + 1e-30f, M = m_up + __builtin_amdgcn_sqrtf(bn) * s * 1.001f;
    row[6] = -(qq * s2);  // (negated: the chain starts from it)
    // th(R) = R^2 k1 + R thr_r + thr_1:  epsR + Ew_l + E1_l  (header comment), coefficients rounded up -- twice: the
    // second (1 + 2^-9), which covers the roundings of th's own evaluation, used to be a multiplication per point
    row[7] = (fmaf(9.86e-4f, Q, 9.7e-7f * M) * kUp + 1e-30f) * kUp;
    row[8] = ((fmaf(9.6e-7f * Q, Q, 2.4e-7f * Q) + fmaf(4.85e-7f * M, M, 1e-12f)) * kUp) * kUp;
}
// th's two cluster-independent coefficients with the same factor inside: 2^-22 |y| (the chain's rounding) and 7.4e-6 R^2
constexpr float kThY = 2.384185791015625e-7f * 1.001953125f, kThR2 = 7.4e-6f * 1.001953125f;

// A operand of a lane's cluster (layout as in filter_assign_body): lower half-wave lanes carry k-slots 0..7 =
// {hi(2 q)_0..5, -|q|^2 + const as hi, lo * 2^10}, upper half-wave lanes k-slots 8..15 = {lo(2 q)_0..5, slope, 0}
__device__ __forceinline__ u32x4 pk_a_operand(const float (&c)[6], float bn, bool valid, const float *hdr, float s, float m_up,
                                              int half) {
    constexpr float kUp = 1.001953125f, kTiny = 1.1920928955078125e-7f;  // (1 + 2^-9) v + 2^-23 survives the rtz to f16
    const float s2 = s * s;
    unsigned ch[3] = {0u, 0u, 0u}, cl[3] = {0u, 0u, 0u};
    float nb = -60000.0f;
    unsigned ebd = 0u;
    if (valid) {
        float ct[6], qq = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            ct[i] = c[i] - hdr[i];
            qq = fmaf(ct[i], ct[i], qq);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) split_f16(ct[2 * p], ct[2 * p + 1], 2.0f * s, ch[p], cl[p]);
        const float Q = sqrtf(qq) * s * 1.001f + 1e-30f, M = m_up + sqrtf(bn) * s * 1.001f;
        // u_j - epsR(R) = t'_j + R ebd_r + ebd_1
        const float ebd_r = fmaf(9.95e-4f, Q, fmaf(9.7e-7f, M, 4.8e-7f));
        // (+ 2e-10: the 2^-34 of E2 and, for a cluster so close to mu that -|q|^2 + const is positive, what the
        // round-toward-zero hi / lo pair below can fall short of it: < 2^-24 / 1024 = 5.8e-11)
        const float ebd_1 = fmaf(5.4e-6f * Q, Q, 4.8e-7f * Q) + fmaf(4.85e-7f * M, M, 2e-10f);
        nb = fmaf(-qq, s2, ebd_1 * kUp);
        nb = fmaf(fabsf(nb), 3.814697265625e-6f, nb) + 1e-12f;  // + 2^-18 |nb|: the hi / lo pair below never rounds it down
        ebd = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(fmaf(ebd_r, kUp, kTiny), 0.f));
    }
    const auto nh = __builtin_amdgcn_cvt_pkrtz(nb, 0.f);
    const unsigned bnd = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(nb, (nb - (float)nh[0]) * 1024.0f));
    return half == 0 ? u32x4{ch[0], ch[1], ch[2], bnd} : u32x4{cl[0], cl[1], cl[2], ebd};
}

// The tables above from the NEW centroids `cen` (d x K, LDS) while the update that made them is still reducing its error and
// flags: role 0 = label table (lanes < K), role 1 = matrix operand of the 64 (col, half) lanes -> tables[lane], role 2 =
// the exact rows for the drains (stage_centroids).  One wavefront per role; nothing written here overlaps the prologue's
// scratch (sC / sL lie behind the accumulator copies).  Speculative: if the update ends the fit or the launch falls back
// to the fp32 filter, the tables are simply not used.
__device__ __forceinline__ void packed_tables_side(int role, int lane, const float *cen, const float *hdr, int K, u32x4 *tables) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *sC = reinterpret_cast<float *>(smem_raw + sizeof(long long) * kPkAccCopies * kPkAccPitch);
    float *sL = sC + K * 8;
    const float s = hdr[6], m_up = hdr[7];
    if (role == 2) {
        if (lane < K) {
            float bn = 0.f;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float v = cen[i * K + lane];
                sC[lane * 8 + i] = v;
                bn = bn + v * v;  // kmeans.py:74 |b|^2: sequential sum of rounded squares (stage_centroids)
            }
            sC[lane * 8 + 6] = bn;
        }
        return;
    }
    const int col = lane & 31, half = lane >> 5;
    const int j = role == 0 ? lane : 2 * (4 * (col >> 3) + (col & 3)) + ((col >> 2) & 1);
    float c[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float bn = 0.f;
    if (j < K) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            c[i] = cen[i * K + j];
            bn = bn + c[i] * c[i];
        }
    }
    if (role == 0) {
        if (j < K) pk_table_row(c, bn, hdr, s, m_up, sL + j * kPkRow);
    } else {
        tables[lane] = pk_a_operand(c, bn, j < K, hdr, s, m_up, half);
    }
}

// (FIRST: the instantiation a fit's first launch takes -- the only one that can meet iteration 0, i.e. the exact scan that
// also writes the packed copy; the launches after it do not carry that code)
template <int NREGS, bool FIRST>
__device__ __forceinline__ void packed_assign_body(const LloydPacked pk, const float *hdr,
                                                   const float *__restrict__ X, int64_t N, int K,
                                                   const et_kmeans_state *state, const float *cen,
                                                   uint8_t *__restrict__ labels, long long *__restrict__ lanes,
                                                   int copy_mask, int range_bad = -1, const u32x4 *tables = nullptr) {
    const unsigned tx = thread_x();  // (opaque per call: see thread_x)
    constexpr int d = 6;
    const int n_thr = (int)blockDim.x, n_wav = n_thr >> 6;
    // (hdr: the caller's copy of *pk.hdr in LDS -- mu[6], s, mu_norm, ok --, requested together with the kernel's other
    // prologue loads: read here, it would be one more dependent round trip to memory in every launch)
    // (everything the decisions below read from LDS is requested at once: read where it is used -- behind one branch after
    // the other -- it was five dependent round trips, ~500 cycles of every launch's prologue)
    const int64_t st_iter = state->iter, st_fast_ok = state->fast_ok;
    const int frac = (int)state->frac;
    const float s = hdr[6], m_up = hdr[7];
    const unsigned pk_ok = __float_as_uint(hdr[8]);
    if (FIRST && st_iter <= 0 && pk.fused) {
        // the fit's first launch: the exact scan of every point -- which also writes the packed copy, from the rows it reads
        // anyway (the scan is bound by its arithmetic, ~130 us at 1e7 points, and has the memory side to spare: a pass of
        // its own over X, kmeans_pack_kernel, costs 175-190 us)
        PackOut po;
        po.xh = const_cast<unsigned *>(pk.xh);
        po.rr = const_cast<unsigned short *>(pk.rr);
        po.xa = const_cast<float4 *>(pk.xa);
        if (!packed_header(X, N, state, const_cast<PackedHeader *>(pk.hdr), po.mu, po.s)) po.xh = nullptr;
        assign_body_valu<6, 4>(X, N, d, K, state, cen, nullptr, labels, nullptr, lanes, copy_mask, po);
        return;
    }
    bool fallback = st_iter <= 0 || !st_fast_ok || pk_ok == 0u;
    fallback = fallback || N > kPkDualMaxN || (N & 3) != 0;  // (the rows are requested through 32-bit buffer offsets, 16 bytes at a time)
    if (!fallback) {  // every |s (c - mu)| inside the packed range?  (cen: d x K floats in LDS, the same in every workgroup)
        if (range_bad >= 0) {  // (the caller's update has looked already: uniform over the workgroup)
            fallback = range_bad != 0;
        } else {
            int bad = 0;
            for (int e = tx; e < d * K; e += n_thr) bad |= !(fabsf((cen[e] - hdr[e / K]) * s) < 31.0f);
            fallback = __syncthreads_or(bad) != 0;
        }
    }
    if (fallback) {
        filter_assign_body<NREGS, false>(X, N, K, state, cen, labels, nullptr, lanes, copy_mask);
        return;
    }
    const int plen = d * K + K + 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    long long *sAcc = reinterpret_cast<long long *>(smem_raw);                                 // kPkAccCopies x kPkAccPitch
    float *sC = reinterpret_cast<float *>(smem_raw + sizeof(long long) * kPkAccCopies * kPkAccPitch);  // K * 8: exact rows (drain)
    float *sL = sC + K * 8;                                                                    // K * kPkRow: label table
    const int lane = tx & 63, wave = tx >> 6, half = lane >> 5, col = lane & 31;
    unsigned *queue = reinterpret_cast<unsigned *>(sL + K * kPkRow) + wave * kPkQueue;
    u32x4 a1;
    __shared__ int sNext;
    if (tables) {
        // the caller's update made the tables beside its reductions (packed_tables_side): sC, sL and the operand are there
        if (tx == 0) sNext = n_wav;
        for (int i = tx; i < kPkAccCopies * kPkAccPitch; i += n_thr) sAcc[i] = 0;  // (the prologue scratch inside it is dead)
        a1 = tables[lane];
    } else {
        stage_centroids(cen, d, K, sC);
        if (tx == 0) sNext = n_wav;
        __syncthreads();  // (`cen` -- the chained kernel's prologue scratch -- lies inside the accumulator copies: cleared only now)
        for (int i = tx; i < kPkAccCopies * kPkAccPitch; i += n_thr) sAcc[i] = 0;
        // per cluster: the centred, scaled row for the old-label chain and the two threshold coefficients
        for (int j = tx; j < K; j += n_thr) {
            float c[d];
#pragma unroll
            for (int i = 0; i < d; ++i) c[i] = sC[j * 8 + i];
            pk_table_row(c, sC[j * 8 + 6], hdr, s, m_up, sL + j * kPkRow);
        }
        {
            const int j = 2 * (4 * (col >> 3) + (col & 3)) + ((col >> 2) & 1);
            float c[d] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            float bn = 0.f;
            if (j < K) {
#pragma unroll
                for (int i = 0; i < d; ++i) c[i] = sC[j * 8 + i];
                bn = sC[j * 8 + 6];
            }
            a1 = pk_a_operand(c, bn, j < K, hdr, s, m_up, half);
        }
    }
    const f16x8 A1 = __builtin_bit_cast(f16x8, a1);
    __syncthreads();  // sL complete
    KM_PSTAMP(5);
    const float4 *l4 = reinterpret_cast<const float4 *>(sL);

    int qn = 0;  // wave-uniform number of queued points
    const int64_t n_groups = (N + 255) / 256;
    const bool dynamic = n_groups > (int64_t)gridDim.x * n_wav;
    bool first = true;
    auto take = [&]() -> int64_t {  // this wavefront's next pass, or -1 (wave-uniform); see filter_assign_body
        int64_t g;
        if (first) {
            first = false;
            g = (int64_t)blockIdx.x * n_wav + wave;
        } else if (dynamic) {
            int i = 0;
            if (lane == 0) i = atomicAdd(&sNext, 1);
            i = __builtin_amdgcn_readfirstlane(i);
            g = (int64_t)blockIdx.x * n_wav + (i % n_wav) + (int64_t)(i / n_wav) * gridDim.x * n_wav;
        } else {
            g = n_groups;
        }
        return g < n_groups ? g : -1;
    };
    PkSrc src;
#pragma unroll
    for (int p = 0; p < 3; ++p) src.row[p] = pk_rsrc(pk.xh + (int64_t)p * N, 4 * N);
    src.rr = pk_rsrc(pk.rr, 2 * N);
    src.lab = pk_rsrc(labels, N);
    const unsigned lane_off = 16u * (unsigned)col, own_off = 128u * (unsigned)half;
    const unsigned ones = 0x14003c00u;  // {1, 2^-10}: partners of {hi, lo * 2^10} of -|q|^2 + const
    // one pass on the rows in `cu` (pass index gc); the other register set is in flight meanwhile
    auto process = [&](const PkRows &cu, int64_t gc) __attribute__((always_inline)) {
        const int64_t n = gc * 256 + 128 * half + 4 * col;
        const bool valid = n < N;
        const unsigned old_packed = cu.lp;
        unsigned undecided = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned w[3];
            u32x4 bLo, bUp;
            const unsigned rpL = cu.rL[q >> 1], rpU = cu.rU[q >> 1];
            // low half: the point's R (the high half meets a zero of the A operand)
            const unsigned r16L = (q & 1) ? (rpL >> 16) : rpL, r16U = (q & 1) ? (rpU >> 16) : rpU;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                bLo[p] = cu.vL[p][q];
                bUp[p] = cu.vU[p][q];
                w[p] = half ? bUp[p] : bLo[p];
            }
            bLo[3] = half ? r16L : ones;
            bUp[3] = half ? r16U : ones;
            const unsigned r16 = half ? r16U : r16L;
            // certified lower bound of the old label's value: fp32 chain on the f16 coordinates
            const int ol = (int)((old_packed >> (8 * q)) & 0xffu);
            const float4 r0 = l4[3 * ol], r1 = l4[3 * ol + 1], r2 = l4[3 * ol + 2];
            float y = r1.z;  // -|q_l|^2 s^2
            y = fma_mix_lo(w[0], r0.x, y);
            y = fma_mix_hi(w[0], r0.y, y);
            y = fma_mix_lo(w[1], r0.z, y);
            y = fma_mix_hi(w[1], r0.w, y);
            y = fma_mix_lo(w[2], r1.x, y);
            y = fma_mix_hi(w[2], r1.y, y);
            const f16x8 BL = __builtin_bit_cast(f16x8, bLo), BU = __builtin_bit_cast(f16x8, bUp);
            f32x16 accL, accU;
#pragma unroll
            for (int r = 0; r < 16; ++r) accL[r] = accU[r] = 0.f;
            accL = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, BL, accL, 0, 0, 0);
            accU = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, BU, accU, 0, 0, 0);
            float bL, sL_, bU, sU;
            top2<NREGS>(accL, bL, sL_);
            top2<NREGS>(accU, bU, sU);
            const auto rb = __builtin_amdgcn_permlane32_swap(__float_as_uint(bL), __float_as_uint(bU), false, false);
            const auto rq = __builtin_amdgcn_permlane32_swap(__float_as_uint(sL_), __float_as_uint(sU), false, false);
            const float b0 = __uint_as_float(rb[0]), b1 = __uint_as_float(rb[1]);
            const float s0 = __uint_as_float(rq[0]), s1 = __uint_as_float(rq[1]);
            const float second = vmed3(b0, b1, vmax(s0, s1));  // second largest upper bound
            // th(R) (1 + 2^-9): the factor is inside the coefficients (pk_table_row: r1.w, r2.x; kThY, kThR2 here)
            const float R = (float)__builtin_bit_cast(_Float16, (unsigned short)(r16 & 0xffffu));
            const float th = fmaf(fabsf(y), kThY, fmaf(R, fmaf(R, kThR2, r1.w), r2.x));
            const bool keep = y - second > th;
            undecided |= (valid && !keep) ? (1u << q) : 0u;
        }
        if (__ballot(undecided != 0u)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool push = (undecided >> q) & 1u;
                const unsigned long long m = __ballot(push);
                if (push) {
                    unsigned *e = queue + qn + __popcll(m & ((1ull << lane) - 1ull));
                    e[0] = (unsigned)(n + q);
                    e[kFilterSlots] = (old_packed >> (8 * q)) & 0xffu;
                }
                qn += __popcll(m);
#ifdef ET_FILTER_DEBUG
                if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long *>(&sAcc[d * K + K + 1]), (unsigned long long)__popcll(m));
#endif
                if (qn >= 64) {
                    qn -= 64;
                    packed_drain(queue + qn, 64, K, sC, pk.xa, labels, sAcc, frac, lane);
                }
            }
        }
    };
    // the loop, unrolled by two over the register sets ra / rb: a set is requested one pass ahead and never copied
    PkRows ra, rb;
#ifdef ET_EXP_WAITSTAMP  // (passes and their cycles only: the exposed wait and the drains are no longer separable)
    unsigned long long ws_wait = 0, ws_pass = 0, ws_drain = 0, ws_n = 0, ws_nd = 0;
#define KM_WS_PASS(call)                                             \
    do {                                                             \
        const unsigned long long t0_ = __builtin_amdgcn_s_memtime(); \
        call;                                                        \
        ws_pass += __builtin_amdgcn_s_memtime() - t0_;               \
        ++ws_n;                                                      \
    } while (0)
#else
#define KM_WS_PASS(call) call
#endif
    int64_t g = take();
    packed_issue_dual(src, g, lane_off, own_off, ra);
    while (g >= 0) {
        const int64_t g2 = take();
        packed_issue_dual(src, g2, lane_off, own_off, rb);
        KM_WS_PASS(process(ra, g));
        if (g2 < 0) break;
        g = take();
        packed_issue_dual(src, g, lane_off, own_off, ra);
        KM_WS_PASS(process(rb, g2));
    }
#undef KM_WS_PASS
#ifdef ET_EXP_WAITSTAMP
    if (lane == 0) {  // (into LDS: six device atomics per wavefront here made the build's launches 4-5x slower)
        atomicAdd(&s_ws_acc[0], ws_n);
        atomicAdd(&s_ws_acc[1], ws_wait);
        atomicAdd(&s_ws_acc[2], ws_pass);
        atomicAdd(&s_ws_acc[3], ws_drain);
        atomicAdd(&s_ws_acc[4], ws_nd);
        atomicAdd(&s_ws_acc[6], 1ull);
    }
#endif
    KM_PSTAMP(6);
    if (qn) packed_drain(queue, qn, K, sC, pk.xa, labels, sAcc, frac, lane);
    __syncthreads();
    KM_PSTAMP(7);
    for (int i = tx; i < plen; i += n_thr) {  // the copies -> copy 0
        long long v = sAcc[i];
#pragma unroll
        for (int c = 1; c < kPkAccCopies; ++c) v += sAcc[c * kPkAccPitch + i];
        sAcc[i] = v;
    }
    __syncthreads();
    emit_partials(sAcc, plen, n_thr, nullptr, lanes, copy_mask);
#ifdef ET_EXP_WAITSTAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    KM_PSTAMP(8);
    KM_PSTAMP_FLUSH();
}

template <int NREGS>
__global__ __launch_bounds__(kFilterMaxThreads) void kmeans_assign_filter_kernel(
    const float *__restrict__ X, int64_t N, int K, const et_kmeans_state *__restrict__ state,
    const float *__restrict__ cen, uint8_t *__restrict__ labels, long long *__restrict__ block_partials,
    long long *__restrict__ lanes) {
    if (state->done) return;
    filter_assign_body<NREGS>(X, N, K, state, cen, labels, block_partials, lanes);
}

// Fold the workgroup deltas into the shard's running totals: one workgroup per entry, unit-stride
// reads.  Cluster sums / counts accumulate across iterations (deltas), the similarity sum and
// the NaN count are per-iteration quantities and are overwritten.
__global__ __launch_bounds__(kKmThreads) void kmeans_reduce_partials_kernel(const long long *__restrict__ block_partials,
                                                                            int n_blocks, int plen, int full,
                                                                            const et_kmeans_state *__restrict__ state,
                                                                            long long *totals, long long *partials) {
    if (state->done) return;
    __shared__ long long sW[kKmThreads / 64];
    const int e = blockIdx.x;
    long long s = 0;
    for (int b = threadIdx.x; b < n_blocks; b += kKmThreads) s += block_partials[(size_t)e * n_blocks + b];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sW[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kKmThreads / 64; ++w) s += sW[w];
        // the running totals stay in the workspace; the caller's buffer receives a copy it may overwrite (all-reduce
        // in place)
        const bool running = !full && state->iter > 0 && e < plen - 2;
        const long long tot = running ? totals[e] + s : s;
        totals[e] = tot;
        partials[e] = tot;
    }
}

// centroid update + convergence scalars from the (all-reduced) exact sums.  One workgroup.
// `partials` may live in global memory or in LDS (flat addressing); `pre` = the state block if the caller has
// already loaded it.
// `last` (may be null): {d*K floats, then one int64 at the next 8-byte boundary} receives the centroids and sim_frac
// the assignment just consumed was made with -- what kmeans_inertia_kernel needs to evaluate its inertia afterwards.
// x[lane + O] for the lanes that are multiples of 2 O (what a level of a "x[i] += x[i + O]" tree needs), without the LDS
// crossbar: inside a row of 16 lanes a DPP row shift, across rows v_permlane16_swap / v_permlane32_swap.  (__shfl_down is a
// ds_bpermute per 32-bit half and ~130 cycles per level; the update's two reduction trees were ~800 cycles of every
// launch's prologue, profiles/r04k_lloyd_launch_stamps.txt.)
template <int O>
__device__ __forceinline__ unsigned lane_down_u32(unsigned v) {
    static_assert(O == 1 || O == 2 || O == 4 || O == 8 || O == 16 || O == 32, "a power of two below the wavefront size");
    if constexpr (O < 16) {
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + O, 0xf, 0xf, true);  // row_shl:O
    } else if constexpr (O == 16) {
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t r = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // second result: rows (1, 1, 3, 3)
        return r.y;
    } else {
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t r = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // second result: (upper half, upper half)
        return r.y;
    }
}
template <int O>
__device__ __forceinline__ double lane_down_f64(double v) {
    const unsigned lo = lane_down_u32<O>((unsigned)__double2loint(v)), hi = lane_down_u32<O>((unsigned)__double2hiint(v));
    return __hiloint2double((int)hi, (int)lo);
}
template <int O>
__device__ __forceinline__ unsigned long long lane_down_u64(unsigned long long v) {
    const unsigned lo = lane_down_u32<O>((unsigned)v), hi = lane_down_u32<O>((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// the minimum of a 64-bit key over the wavefront, in lane 0 (register exchanges only)
__device__ __forceinline__ unsigned long long wave_min_u64_lane0(unsigned long long key) {
    unsigned long long o;
    o = lane_down_u64<32>(key); key = o < key ? o : key;
    o = lane_down_u64<16>(key); key = o < key ? o : key;
    o = lane_down_u64<8>(key); key = o < key ? o : key;
    o = lane_down_u64<4>(key); key = o < key ? o : key;
    o = lane_down_u64<2>(key); key = o < key ? o : key;
    o = lane_down_u64<1>(key); key = o < key ? o : key;
    return key;
}

struct NoSideWork {
    __device__ __forceinline__ void operator()(int) const {}
};
// `side(w)`: work for wavefront 2 + w of the workgroup, run beside the reductions (between the update's two barriers) --
// it may read the new centroids in `cen`
template <class Side = NoSideWork>
__device__ __forceinline__ void update_body(et_kmeans_state *state, const long long *partials, int d, int K, float tol,
                                            float *cen, float *trace, const et_kmeans_state *pre = nullptr,
                                            float *last = nullptr, bool need_inertia = true,
                                            const float *pk_hdr = nullptr, int *pk_bad = nullptr, Side side = Side()) {
    const unsigned tx = thread_x();  // (opaque per call: see thread_x)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *sSq = reinterpret_cast<float *>(smem_raw);  // d*K squared differences
    float *sNew = sSq + d * K;
    // every global value the serial tail needs is fetched up front (one round trip instead of a chain of them)
    const et_kmeans_state st = pre ? *pre : *state;
    const long long sim_sum = partials[d * K + K], nan_count = partials[d * K + K + 1];
    const int frac = (int)st.frac;
    const double inv_scale = ldexp(1.0, -frac);
    for (int e = tx; e < d * K; e += (int)blockDim.x) {
        const int j = e % K;
        const long long cnt = partials[d * K + j];
        float c;
        if (cnt == 0) c = __int_as_float(0x7fc00000);  // 0/0 (kmeans.py:182)
        else c = (float)(((double)partials[e] * inv_scale) / (double)cnt);
        const float prev = cen[e];
        const float diff = prev - c;  // kmeans.py:48
        sSq[e] = diff * diff;         // :49
        sNew[e] = c;
        cen[e] = c;
        if (last) last[e] = prev;
    }
    if (last && tx == 0) *reinterpret_cast<long long *>(last + ((d * K + 1) & ~1)) = (long long)st.sim_frac;
    __syncthreads();
    // max |c| (NaN ignored), smallest non-zero |c| and a non-finite flag, reduced by the first wavefront
    __shared__ float sRed[3];
    __shared__ double sErr;
    if (blockDim.x <= 64 || (tx >> 6) == 1) {
        // kmeans.py:50 in the oracle's fixed order (oracle/et_oracle.c: eto_error_sum): fp64, blocks of 256 consecutive
        // terms, each a balanced tree x[i] += x[i + s], s = 1 ... 128, block results added in block order.  A lane holds
        // four consecutive terms (levels s = 1, 2), the lanes combine through shuffles (s = 4 ... 128): seven dependent
        // additions instead of the d K of a running sum (1.7 us of every Lloyd launch's prologue with d K = 120).
        // It runs on the second wavefront next to the reductions below.
        const int l = tx & 63, dk = d * K;
        double total = 0.0;
        for (int b0 = 0; b0 < dk; b0 += 256) {
            float f[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) f[q] = b0 + 4 * l + q < dk ? sSq[b0 + 4 * l + q] : 0.f;
            double t = ((double)f[0] + (double)f[1]) + ((double)f[2] + (double)f[3]);
            // t += t[lane + o], o = 1 ... 32: valid in the lanes that are multiples of 2 o
            t = t + lane_down_f64<1>(t);
            t = t + lane_down_f64<2>(t);
            t = t + lane_down_f64<4>(t);
            t = t + lane_down_f64<8>(t);
            t = t + lane_down_f64<16>(t);
            t = t + lane_down_f64<32>(t);
            total = total + t;
        }
        if (l == 0) sErr = total;
    }
    if (tx >= 128) side((int)(tx >> 6) - 2);
    if (tx < 64) {
        float mx = 0.f;
        unsigned mn = 0x7f800000u;
        int bad = 0;
        // (d K <= 192 for the shapes the chained kernel takes: up to three values per lane, requested from LDS together with
        // the packed copy's header -- a loop with a dependent header read per value was 1 000 of this phase's 1 600 cycles)
        float h[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (pk_hdr) {
#pragma unroll
            for (int q = 0; q < 7; ++q) h[q] = pk_hdr[q];
        }
        for (int e0 = tx; e0 < d * K; e0 += 3 * 64) {
            float v[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) v[u] = sNew[e0 + 64 * u < d * K ? e0 + 64 * u : e0];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int e = e0 + 64 * u;
                if (e >= d * K) break;
                const float a = fabsf(v[u]);
                if (!(a <= 3.402823466e+38f)) bad = 1;
                if (a > mx) mx = a;  // false for NaN: ignored, like the oracle
                const unsigned b = (unsigned)__float_as_int(a);
                if (a <= 3.402823466e+38f && b != 0u && b < mn) mn = b;
                if (pk_hdr) {  // packed_assign_body's range test of the new centroids (bit 1 of `bad`), while they are at hand
                    float mu = h[0];
#pragma unroll
                    for (int q = 1; q < 6; ++q) mu = (q < d && e >= q * K) ? h[q] : mu;  // mu[e / K]
                    if (!(fabsf((v[u] - mu) * h[6]) < 31.0f)) bad |= 2;
                }
            }
        }
        // (order independent: any tree; lane 0 ends up with the result)
#define ET_DOWN(O)                                                                        \
    do {                                                                                  \
        mx = fmaxf(mx, __uint_as_float(lane_down_u32<O>(__float_as_uint(mx))));           \
        const unsigned other = lane_down_u32<O>(mn);                                      \
        mn = other < mn ? other : mn;                                                     \
        bad |= (int)lane_down_u32<O>((unsigned)bad);                                      \
    } while (0)
        ET_DOWN(32);
        ET_DOWN(16);
        ET_DOWN(8);
        ET_DOWN(4);
        ET_DOWN(2);
        ET_DOWN(1);
#undef ET_DOWN
        if (tx == 0) {
            sRed[0] = mx;
            sRed[1] = __int_as_float((int)mn);
            sRed[2] = (bad & 1) ? 1.f : 0.f;
            if (pk_bad) *pk_bad = bad >> 1;
        }
    }
    __syncthreads();
    if (tx == 0) {
        const float error = (float)sErr;
        const int64_t n_total = st.n_total;
        // (need_inertia == false: a trace-less fit's launches, which do not accumulate the similarity sum -- the inertia of
        // the last assignment is evaluated after the loop -- and the workgroups that publish nothing: an fp64 division
        // less on the serial tail of every launch's prologue)
        float inertia = (float)st.inertia;
        if (need_inertia) {
            if (nan_count > 0) inertia = __int_as_float(0x7fc00000);
            else inertia = (float)(-(((double)sim_sum * ldexp(1.0, -(int)st.sim_frac)) / (double)n_total));  // :57
        }
        const double mc = (double)sRed[0];
        const double mx = st.max_abs_x;
        state->max_abs_c = mc;
        state->sim_frac = sim_frac_bits(mx, mc, d, n_total);
        int64_t fast = 0;
        if (sRed[2] == 0.f && mx < 1e18 && mc < 1e18) {
            const unsigned lim = 0x26800000u;  // 2^-50, see the fast_ok levels above
            fast = ((unsigned)__float_as_int(sRed[1]) >= lim && (unsigned long long)st.min_nz_x_bits >= lim) ? 2 : 1;
        }
        state->fast_ok = fast;
        if (trace) {
            trace[2 * st.iter] = error;
            trace[2 * st.iter + 1] = inertia;
        }
        state->error = (double)error;
        state->inertia = (double)inertia;
        state->iter = st.iter + 1;
        state->done = (error <= tol) ? 1 : 0;  // kmeans.py:239 (NaN -> keep going)
    }
}

__global__ __launch_bounds__(kKmThreads) void kmeans_update_kernel(et_kmeans_state *state,
                                                                   const long long *__restrict__ partials, int d, int K,
                                                                   float tol, float *__restrict__ cen,
                                                                   float *__restrict__ trace) {
    if (state->done) return;
    update_body(state, partials, d, K, tol, cen, trace);
}

// BatchKMeans.fit on a batch of l > 1 problems (kmeans.py:228-240): ONE error -- the squared centroid movement summed
// over all problems (kmeans.py:45-51 on the (l, d, K) tensors) -- is compared with the tolerance and all problems stop
// together.  The step API runs the problems side by side with a tolerance no error can meet; this kernel, after their
// updates, sums the per-problem errors (fp64, problem order; each is the fp32 value the update stored) and sets every
// problem's convergence flag from the sum.  One wavefront.
__global__ void kmeans_joint_done_kernel(et_kmeans_state *const *__restrict__ states, int n, float tol) {
    if (threadIdx.x != 0) return;
    if (states[0]->done) return;  // (the flags are only ever set together)
    double sum = 0.0;
    for (int b = 0; b < n; ++b) sum += states[b]->error;
    const int64_t done = ((float)sum <= tol) ? 1 : 0;  // kmeans.py:239 (NaN -> keep going)
    for (int b = 0; b < n; ++b) states[b]->done = done;
}

// Large shards, single-GPU fit: ONE launch per Lloyd iteration and NO serial section between two iterations.
//
// A launch first applies the update of the PREVIOUS iteration's assignment and then makes its own assignment:
// every workgroup folds the 16 copies of the exact integer totals the previous launch's workgroups added their
// deltas onto, and computes the new centroids, error and convergence flag ITSELF, straight into its LDS staging --
// identical integers in, identical results in every workgroup, so nobody waits for a "last" workgroup (the ticket +
// fence + one-workgroup fold and update + dispatch gap of the two-phase form cost ~9 of the ~20 us that an iteration
// takes besides streaming the points).  Workgroup 0 also publishes the results (state, centroids, totals, trace,
// the centroids of the last assignment).  Nothing a workgroup reads is written during the same launch:
//   state / centroids / totals   two copies, launch t reads copy t % 2 and workgroup 0 writes copy (t+1) % 2
//   the 16-copy delta table       three copies: launch t reads t % 3 (filled by launch t-1), adds onto (t+1) % 3 and
//                                workgroup 0 clears (t+2) % 3 (read by launch t-1, to be filled by launch t+1)
// The assignment of the final iteration is followed by kmeans_chain_finalize_kernel (its update, once).
struct LloydChain {
    const et_kmeans_state *st_rd;
    et_kmeans_state *st_wr;
    const float *cen_rd;
    float *cen_wr;
    const long long *tot_rd;
    long long *tot_wr;
    const long long *lanes_rd;
    long long *lanes_wr;
    long long *lanes_zero;
    float *last;
    unsigned long long *mail;  // host-visible progress word (et_hostring.h: mailbox), or nullptr
    // sharded loop: ONE copy of the delta table, entries adjacent (what travels over the wire between two launches is
    // then the d K + K + 2 int64 that carry the information, 1.1 KB, not the 16-copy table)
    int compact;
    int copies;  // compact copies of the delta table in use (1: sharded loop -- the wire format; a power of two <= 8 else)
    int vec_ok;  // this shard's rows allow 16-byte loads (N % 4 == 0, aligned) and it has >= 1024 points: filter body
    LloydPacked pk;  // pk.xh != nullptr: trace-less iterations run on the packed copy (packed_assign_body)
};

// fold the 16 copies of every total (layout [entry][copy]: a linear, coalesced sweep, 16 adjacent lanes per entry) and
// add the running totals of the earlier iterations -> sTot (LDS)
// The fold of the kAccLanes copies of the delta table onto the previous totals, in two halves so that the caller can
// put its other loads between them: fold_issue() requests every value (clamped indices keep the register arrays out
// of scratch memory), fold_combine() sums.  d = 6, K <= 32 with the filter kernels' 768 / 1024 threads needs 3 ... 5
// sweeps of blockDim.x entries; fold_lanes() is the plain loop for any other shape.
constexpr int kTimedRun = 4;  // launches between the two events of a timed sample of the chained loop
constexpr int kFoldSweeps = 5;
struct FoldRegs {
    long long v[kFoldSweeps], prev[kFoldSweeps];
};
__device__ __forceinline__ bool fold_fits(int plen) { return kFoldSweeps * (int)blockDim.x >= plen * kAccLanes; }
__device__ __forceinline__ void fold_issue(const long long *__restrict__ lanes, const long long *__restrict__ tot_prev,
                                           int plen, FoldRegs &r, bool compact = false, int copies = 1) {
    const int total = plen * kAccLanes, n_threads = (int)blockDim.x;
    if (compact) {  // entries adjacent: one load per entry and copy (plen <= 226 <= blockDim.x), all requested together
        const int ci = (int)threadIdx.x < plen ? (int)threadIdx.x : 0;
        const int pitch = compact_pitch(plen);
        long long v = lanes[ci];
        long long x[3] = {0, 0, 0};
        if (copies > 1) x[0] = lanes[ci + pitch];
        if (copies > 2) {
            x[1] = lanes[ci + 2 * pitch];
            x[2] = lanes[ci + 3 * pitch];
        }
        long long y = 0;
        for (int c = 4; c < copies; ++c) y += lanes[ci + c * pitch];
        r.v[0] = ((v + x[0]) + (x[1] + x[2])) + y;  // (integers: any order)
        r.prev[0] = tot_prev[ci];
        return;
    }
#pragma unroll
    for (int it = 0; it < kFoldSweeps; ++it) {
        const int idx = it * n_threads + (int)threadIdx.x;
        const int ci = idx < total ? idx : 0;
        r.v[it] = lanes[ci];
        r.prev[it] = tot_prev[ci / kAccLanes];
    }
}
__device__ __forceinline__ void fold_combine(const FoldRegs &r, bool have_prev, int plen, long long *sTot,
                                             bool compact = false) {
    const int total = plen * kAccLanes, n_threads = (int)blockDim.x;
    if (compact) {
        const int e = (int)threadIdx.x;
        if (e < plen) sTot[e] = ((have_prev && e < plen - 2) ? r.prev[0] : 0) + r.v[0];
        return;
    }
#pragma unroll
    for (int it = 0; it < kFoldSweeps; ++it) {
        const int idx = it * n_threads + (int)threadIdx.x;
        if (it * n_threads >= total) break;  // uniform
        long long x = idx < total ? r.v[it] : 0;
#pragma unroll
        for (int o = kAccLanes / 2; o > 0; o >>= 1) x += __shfl_xor(x, o);
        if (idx < total && (idx & (kAccLanes - 1)) == 0) {
            const int e = idx / kAccLanes;
            sTot[e] = ((have_prev && e < plen - 2) ? r.prev[it] : 0) + x;
        }
    }
}
__device__ __forceinline__ void fold_lanes(const long long *__restrict__ lanes, const long long *__restrict__ tot_prev,
                                           bool have_prev, int plen, long long *sTot, bool compact = false, int copies = 1) {
    if (compact) {
        for (int e = threadIdx.x; e < plen; e += (int)blockDim.x) {
            long long v = lanes[e];
            for (int c = 1; c < copies; ++c) v += lanes[e + c * compact_pitch(plen)];
            sTot[e] = ((have_prev && e < plen - 2) ? tot_prev[e] : 0) + v;
        }
        return;
    }
    if (fold_fits(plen)) {
        FoldRegs r;
        fold_issue(lanes, tot_prev, plen, r);
        fold_combine(r, have_prev, plen, sTot);
        return;
    }
    const int total = plen * kAccLanes, n_threads = (int)blockDim.x;
    for (int base = 0; base < total; base += n_threads) {
        const int idx = base + (int)threadIdx.x;
        long long v = idx < total ? lanes[idx] : 0;
#pragma unroll
        for (int o = kAccLanes / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (idx < total && (idx & (kAccLanes - 1)) == 0) {
            const int e = idx / kAccLanes;
            sTot[e] = ((have_prev && e < plen - 2) ? tot_prev[e] : 0) + v;
        }
    }
}

template <int NREGS, bool SIM, bool FIRST = false>
__global__ __launch_bounds__(kFilterMaxThreads) void kmeans_lloyd_chain_kernel(
    const float *__restrict__ X, int64_t N, int K, const LloydChain ch, uint8_t *__restrict__ labels, float tol,
    float *trace, int has_pending) {
    constexpr int d = 6;
    const int plen = d * K + K + 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ et_kmeans_state sSt;
    // scratch of this prologue, inside the area the assignment's LDS queues use later: the folded totals past the
    // 2 d K floats update_body works in, then the centroids (old -> new, in place)
    long long *sTot = reinterpret_cast<long long *>(smem_raw) + 512;
    float *sCen = reinterpret_cast<float *>(sTot + ((plen + 1) & ~1));
    const bool wg0 = blockIdx.x == 0;
    // (no local copy of the state block: a by-value et_kmeans_state whose address is taken ends up in scratch memory,
    // and a kernel with a private segment pays for it at every wavefront launch)
    // Everything the prologue needs from memory is requested at once -- the convergence flag, the centroids (d K <= 192
    // <= blockDim.x values), the delta table and the previous totals: one round trip, not three dependent ones.
    KM_PSTAMP(0);
    const int64_t done0 = ch.st_rd->done, iter0 = ch.st_rd->iter;
    __shared__ float sPkHdr[12];
    __shared__ u32x4 sPkTab[64];  // packed_assign_body's matrix operand per lane, when made beside the update
    bool tables_ready = false;
    __shared__ int sPkBad;  // the packed copy's range test of the new centroids, made by update_body (-1: not made)
    if (threadIdx.x == 0) sPkBad = -1;
    float pk_word = 0.f;
    if constexpr (!SIM) {  // the packed copy's header (9 words), with the other prologue loads
        if (ch.pk.xh && threadIdx.x < 9) pk_word = reinterpret_cast<const float *>(ch.pk.hdr)[threadIdx.x];
    }
    constexpr int kStateWords = (int)(sizeof(et_kmeans_state) / sizeof(unsigned));
    const unsigned st_word = reinterpret_cast<const unsigned *>(ch.st_rd)[(int)threadIdx.x < kStateWords ? (int)threadIdx.x : 0];
    const float cen0 = ch.cen_rd[(int)threadIdx.x < d * K ? (int)threadIdx.x : 0];
    FoldRegs fr;
    // (the delta table of the chained loop is ALWAYS the compact one-copy form -- host side, chain_for() --: a constant here,
    // so that the sweeps of the 16-copy form are not compiled in; their register arrays, indexed under a runtime flag, ended
    // up in scratch memory: a store -> load round trip in every launch's prologue and a private segment per wavefront)
    fold_issue(ch.lanes_rd, ch.tot_rd, plen, fr, true, ch.copies);
    if (done0) {  // converged earlier (or bad input flagged before the loop): keep the published copies in step
        if (wg0) {
            if (threadIdx.x == 0) {
                *ch.st_wr = *ch.st_rd;
                if (ch.mail)  // the host stops launching as soon as it reads the flag (it would otherwise spin for it)
                    __hip_atomic_store(ch.mail, (1ull << 63) | (unsigned long long)iter0, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
            }
            for (int e = threadIdx.x; e < d * K; e += (int)blockDim.x) ch.cen_wr[e] = ch.cen_rd[e];
            for (int e = threadIdx.x; e < plen; e += (int)blockDim.x) ch.tot_wr[e] = ch.tot_rd[e];
        }
        return;
    }
#ifdef ET_EXP_WAITSTAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    KM_PSTAMP(1);
    if ((int)threadIdx.x < d * K) sCen[threadIdx.x] = cen0;
    if (!SIM && threadIdx.x < 9) sPkHdr[threadIdx.x] = pk_word;
    if ((int)threadIdx.x < kStateWords) reinterpret_cast<unsigned *>(&sSt)[threadIdx.x] = st_word;  // the state block, word by word
    if (has_pending) {
        fold_combine(fr, iter0 > 0, plen, sTot, true);
        __syncthreads();
        KM_PSTAMP(2);
        // (trace-less fit on the packed copy: the next assignment's tables are made by wavefronts 2, 3 and 4 beside the
        // update's reductions -- 1.3 us of every launch's prologue when they followed it)
        const bool side_tables = !SIM && ch.pk.xh && ch.vec_ok && blockDim.x >= 320;
        auto side = [&](int w) {
            if (side_tables && w < 3) packed_tables_side(w, (int)(threadIdx.x & 63), sCen, sPkHdr, K, sPkTab);
        };
        update_body(&sSt, sTot, d, K, tol, sCen, wg0 ? trace : nullptr, nullptr, wg0 ? ch.last : nullptr, SIM && wg0,
                    (!SIM && ch.pk.xh) ? sPkHdr : nullptr, &sPkBad, side);  // reads its copy in LDS
        tables_ready = side_tables;
        KM_PSTAMP(3);
    }
    __syncthreads();
    const int64_t done1 = sSt.done;  // (requested together: the flag of the update just applied and its range test)
    const int range_bad = sPkBad;
    if (wg0) {  // publish (read by the next launch, the host's convergence polling and the finalize kernel)
        if (threadIdx.x == 0) {
            *ch.st_wr = sSt;
            if (ch.mail)
                __hip_atomic_store(ch.mail, ((unsigned long long)(sSt.done != 0) << 63) | (unsigned long long)sSt.iter,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        for (int e = threadIdx.x; e < d * K; e += (int)blockDim.x) ch.cen_wr[e] = sCen[e];
        if (has_pending)
            for (int e = threadIdx.x; e < plen; e += (int)blockDim.x) ch.tot_wr[e] = sTot[e];
        for (int i = threadIdx.x; i < ch.copies * compact_pitch(plen); i += (int)blockDim.x) ch.lanes_zero[i] = 0;  // (the compact copies in use)
        // sCen / sTot lie inside the area the assignment bodies clear for their accumulators: this workgroup's other
        // wavefronts must not start clearing while the ones above still read (uniform per workgroup: only workgroup 0 waits)
        __syncthreads();
    }
    KM_PSTAMP(4);
    if (done1) return;  // the update just applied met the tolerance: no further assignment (kmeans.py:239)
    const int copy_mask = -ch.copies;  // (-1: one compact copy)
    // a shard whose rows do not allow 16-byte loads (sharded runs cut the points anywhere), or a tiny one: the plain exact
    // scan, one point per lane, inside the same launch -- which loop form a sharded fit takes then depends on (d, K)
    // alone and every rank knows it without asking the others
    if (ch.vec_ok) {
        if constexpr (!SIM) {
            if (ch.pk.xh) {
                packed_assign_body<NREGS, FIRST>(ch.pk, sPkHdr, X, N, K, &sSt, sCen, labels, ch.lanes_wr, copy_mask, range_bad,
                                          tables_ready ? sPkTab : nullptr);
                return;
            }
        }
        filter_assign_body<NREGS, SIM>(X, N, K, &sSt, sCen, labels, nullptr, ch.lanes_wr, copy_mask);
    } else {
        assign_body_valu<6, 1>(X, N, d, K, &sSt, sCen, nullptr, labels, nullptr, ch.lanes_wr, copy_mask);
    }
}

// After the loop: the update that belongs to the last assignment (if one is pending), into the caller's buffers.
__global__ __launch_bounds__(kKmThreads) void kmeans_chain_finalize_kernel(const LloydChain ch, et_kmeans_state *state,
                                                                           long long *partials, float *cen, int d, int K,
                                                                           float tol, float *trace, int has_pending,
                                                                           long long *sim_total, int last_was_sim = 0) {
    if (sim_total && threadIdx.x < 2) sim_total[threadIdx.x] = 0;  // for the inertia pass that follows a trace-less fit
    // sim_total[2]: the pending assignment was made by a launch that accumulated the similarity sum (the trace-less loop's
    // LAST launch when it runs to max_iter) -- the update below turns it into the inertia and the inertia pass is skipped
    if (sim_total && threadIdx.x == 2) sim_total[2] = (last_was_sim && has_pending && !ch.st_rd->done) ? 1 : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int plen = d * K + K + 2;
    long long *sTot = reinterpret_cast<long long *>(smem_raw) + 512;
    for (int e = threadIdx.x; e < d * K; e += kKmThreads) cen[e] = ch.cen_rd[e];
    if (ch.st_rd->done || !has_pending) {
        if (threadIdx.x == 0) *state = *ch.st_rd;
        for (int e = threadIdx.x; e < plen; e += kKmThreads) partials[e] = ch.tot_rd[e];
        return;
    }
    fold_lanes(ch.lanes_rd, ch.tot_rd, ch.st_rd->iter > 0, plen, sTot, ch.compact != 0, ch.copies);
    __syncthreads();
    for (int e = threadIdx.x; e < plen; e += kKmThreads) partials[e] = sTot[e];
    if (threadIdx.x == 0) *state = *ch.st_rd;
    __syncthreads();
    update_body(state, sTot, d, K, tol, cen, trace, ch.st_rd, ch.last);
}

// ------------------------------------------------------------------------------------------
// Single-GPU fit, shards the filter takes: ALL Lloyd iterations in ONE launch (persistent workgroups).
//
// What a kernel boundary costs between two iterations, measured on this chip (tools/exp_overlap*.hip,
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
// NaN first) encoded as a 64-bit key so that a plain unsigned min is the reduction.
// ------------------------------------------------------------------------------------------
//
// Steps >= 2 skip the coordinate read of every point that provably keeps its running maximum (Elkan's
// triangle inequality, made rigorous for the computed fp32 similarities): with l = nearest[n] the centroid that
// holds best[n], Delta_l <= ||c_new - c_l|| and E >= the rounding error of any computed similarity of this shard
// (2^-19 (R + C)^2 with R = sqrt(d) max|x| >= every local ||x|| and C = the largest centroid norm so far),
//     ||x - c_l|| <= sqrt(E - best[n])   and   ||x - c_new|| >= Delta_l - ||x - c_l||,
// so  Delta_l >= 2 sqrt(E - best[n])  implies  y_new <= -||x - c_new||^2 + E <= best[n]:  the strict `>` of the
// update cannot fire and best / nearest stay as they are.  Such a point costs 5 B (best + nearest) instead of
// 32 B; farthest-first picks are far from everything by construction, so most points qualify.  best[] is only
// written when it changes.  max|x| is collected by step 1, which reads everything anyway.
// PERSIST (not instantiated any more: tools/lost_forms/kmeans_init_persist.hip.txt): the body inside ONE launch for all steps,
// separated by a fence-free grid barrier: everything that
// crosses workgroups inside the launch -- the workgroup keys, the centroid columns workgroup 0 stores -- is then written
// and read with device-scope atomics (served by the memory side: no cache fence); best / nearest / the tile summaries
// are only re-read by the wavefront that wrote them (the tile -> wavefront map is fixed).
template <int D, bool PERSIST>
__device__ __forceinline__ void init_step_body(const float *__restrict__ X, int64_t N, int d_rt, int K, int step,
                                               const float *C0, float *__restrict__ best, uint8_t *__restrict__ nearest,
                                               unsigned *__restrict__ max_abs_bits, int64_t index_base,
                                               unsigned long long *block_keys, const unsigned long long *prev_keys,
                                               int n_prev, float *C0_rw, unsigned char *cand, uint4 *__restrict__ meta,
                                               int meta_valid) {
    const int d = D ? D : d_rt;
    __shared__ float sc[ET_KMEANS_MAX_D + 1];
    __shared__ float sDelta[ET_KMEANS_MAX_CLUSTERS + 1];
    __shared__ unsigned long long sKey[kKmThreads / 64];
    __shared__ unsigned sMax[kKmThreads / 64];
    __shared__ unsigned sCmax;  // fp32 bits of the largest centroid norm among columns 0 .. step-1
    // the earlier centroids this thread will measure the new one against (columns < step - 1 are final): requested now,
    // so that their round trip overlaps the key reduction and the gather of the new centroid
    float cprev[D ? D : 1];
    if constexpr (D != 0) {
        const int jc = (int)threadIdx.x < step - 1 ? (int)threadIdx.x : 0;
#pragma unroll
        for (int i = 0; i < D; ++i)
            cprev[i] = PERSIST ? __hip_atomic_load(&C0[i * K + jc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : C0[i * K + jc];
    }
    // the tile summaries of this wavefront's first 64 tiles do not depend on the new centroid either: requested now, their
    // round trip (the fourth dependent one of a step) runs under the prologue's
    const int lane = (int)(threadIdx.x & 63);
    const bool vec = step > 1 && ((reinterpret_cast<uintptr_t>(best) & 15u) == 0) && ((reinterpret_cast<uintptr_t>(nearest) & 3u) == 0);
    const int64_t n4 = vec ? N / 4 : 0;
    const int64_t n_tiles = (n4 + 63) >> 6, n_waves = (int64_t)gridDim.x * (kKmThreads / 64);
    const int64_t per = (n_tiles + n_waves - 1) / n_waves;
    const int64_t t_begin = ((int64_t)blockIdx.x * (kKmThreads / 64) + (threadIdx.x >> 6)) * per;
    const int64_t t_end = t_begin + per < n_tiles ? t_begin + per : n_tiles;
    uint4 meta0 = make_uint4(0u, 0u, 0u, 0u);
    if (meta && vec && meta_valid && t_begin + lane < t_end) meta0 = meta[t_begin + lane];
    if (prev_keys) {
        // Single-GPU path: centroid step-1 has not been stored yet -- every workgroup derives it from the previous
        // step's workgroup keys (the same minimum everywhere), workgroup 0 also stores it.  Two short round trips
        // in the prologue instead of a pick launch between two steps.
        unsigned long long key = ~0ull;
        for (int b0 = 0; b0 < n_prev; b0 += 4 * kKmThreads) {  // four keys per thread in flight (usually all there are)
            unsigned long long k4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = b0 + u * kKmThreads + (int)threadIdx.x;
                k4[u] = PERSIST ? __hip_atomic_load(&prev_keys[b < n_prev ? b : 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                : prev_keys[b < n_prev ? b : 0];
                if (b >= n_prev) k4[u] = ~0ull;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) key = k4[u] < key ? k4[u] : key;
        }
        key = wave_min_u64_lane0(key);  // (register exchanges: six ds_bpermute levels on a 64-bit key were ~800 cycles)
        if ((threadIdx.x & 63) == 0) sKey[threadIdx.x >> 6] = key;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < kKmThreads / 64; ++w) key = sKey[w] < key ? sKey[w] : key;
            const int64_t local = (int64_t)(unsigned)(key & 0xffffffffull) - index_base;
            sCmax = 0u;
            float bn = 0.f;
            const bool ok = key != ~0ull && local >= 0 && local < N;
            float pt[D ? D : 1];
            if constexpr (D != 0) {  // the winner's coordinates: all loads first (the stores below may alias for the compiler)
#pragma unroll
                for (int i = 0; i < D; ++i) pt[i] = X[(int64_t)i * N + (ok ? local : 0)];
            }
#pragma unroll
            for (int i = 0; i < (D ? D : ET_KMEANS_MAX_D); ++i) {
                if (i >= d) break;
                float v;
                if constexpr (D != 0) v = ok ? pt[i] : __int_as_float(0x7fc00000);
                else v = ok ? X[(int64_t)i * N + local] : __int_as_float(0x7fc00000);
                sc[i] = v;
                bn = bn + v * v;
                if (blockIdx.x == 0) {
                    if (PERSIST) __hip_atomic_store(&C0_rw[i * K + (step - 1)], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else C0_rw[i * K + (step - 1)] = v;
                    reinterpret_cast<float *>(cand + 8)[i] = v;
                }
            }
            sc[d] = bn;
            if (blockIdx.x == 0) *reinterpret_cast<unsigned long long *>(cand) = key;
        }
    } else if (threadIdx.x == 0) {
        sCmax = 0u;
        float bn = 0.f;
        for (int i = 0; i < d; ++i) {
            const float v = C0[i * K + (step - 1)];
            sc[i] = v;
            bn = bn + v * v;
        }
        sc[d] = bn;
    }
    __syncthreads();
    // lower bounds of the distances from the new centroid to the earlier ones, upper bound of the centroid norms
    for (int j = threadIdx.x; j < step; j += kKmThreads) {
        double s2 = 0.0, n2 = 0.0;
#pragma unroll
        for (int i = 0; i < (D ? D : ET_KMEANS_MAX_D); ++i) {
            if (i >= d) break;
            double cj;
            if constexpr (D != 0) cj = j == step - 1 ? (double)sc[i] : (double)cprev[i];  // step <= K < blockDim.x: j == threadIdx.x
            else cj = j == step - 1 ? (double)sc[i]
                                    : (double)(PERSIST ? __hip_atomic_load(&C0[i * K + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                       : C0[i * K + j]);
            const double t = (double)sc[i] - cj;
            s2 += t * t;
            n2 += cj * cj;
        }
        // the SQUARE of a lower bound of the distance: the skip test below is delta >= 2 sqrt(E - b), evaluated as
        // delta^2 >= 4 (E - b) with E - b >= 0 (no square root per point; both sides carry their margins)
        sDelta[j] = (float)(s2 * (1.0 - 4e-6)) * (1.0f - 1e-6f);
        const float nj = (float)(sqrt(n2) * (1.0 + 1e-6)) * (1.0f + 1e-6f);
        atomicMax(&sCmax, nj == nj ? __float_as_uint(nj) : 0x7f800000u);  // NaN centroid: +inf, nothing is skipped
    }
    __syncthreads();
    float c[D ? D : ET_KMEANS_MAX_D];
#pragma unroll
    for (int i = 0; i < (D ? D : ET_KMEANS_MAX_D); ++i)
        if (i < d) c[i] = sc[i];
    const float bn = sc[d];
    // E: bound on |computed similarity - (-||x - c||^2)| for this shard's points; +inf (never skip) if unknown
    float E = __int_as_float(0x7f800000);
    if (step > 1) {
        const float R = sqrtf((float)d) * __uint_as_float(*max_abs_bits) * 1.0001f + __uint_as_float(sCmax);
        E = R * R * 1.9073486328125e-6f;  // 2^-19 (R + C)^2
        if (!(E <= 3.0e38f)) E = __int_as_float(0x7f800000);
    }
    unsigned long long key = ~0ull;
    float mabs = 0.f;
    // one point: full evaluation unless `skip`; returns the (possibly updated) running maximum
    auto visit = [&](int64_t n, float b, bool skip, float &b_out, int &lab_out) {
        if (!skip) {
            float an = 0.f, y = 0.f;
#pragma unroll
            for (int i = 0; i < (D ? D : ET_KMEANS_MAX_D); ++i)
                if (i < d) {
                    const float v = X[(int64_t)i * N + n];
                    an = an + v * v;
                    y = fmaf(v, c[i], y);
                    if (step == 1) mabs = fmaxf(mabs, fabsf(v));  // NaN ignored; a NaN point never gets skipped anyway
                }
            y = y * 2.0f;
            y = y - an;
            y = y - bn;
            if (step == 1 || gt_nanmax(y, b)) {
                b = y;
                best[n] = b;
                nearest[n] = (uint8_t)(step - 1);
                lab_out = step - 1;
            }
        }
        b_out = b;
        const unsigned long long k = ((unsigned long long)orderable(b) << 32) | (unsigned)(index_base + n);
        key = k < key ? k : key;
    };
    const int64_t stride = (int64_t)gridDim.x * kKmThreads;
    const int64_t tid = (int64_t)blockIdx.x * kKmThreads + threadIdx.x;
    // steps >= 2 look at four points per lane through one 16-B load of best[] and one 4-B load of nearest[]
    // Steps >= 3 first look at a 16-byte summary of each tile of 256 points (the smallest key, the largest running
    // similarity, the set of nearest centroids -- written by the step before): if the skip test holds for the tile's
    // WORST values it holds for every point in it (E - b and the product are monotone in b, the distance bound is the
    // smallest over the labels present), nothing in the tile changes, and its smallest key is the stored one -- the tile
    // costs 16 bytes instead of 1280.  A wavefront owns a contiguous run of tiles; its LANES test up to 64 of them at once
    // (one summary each: one round trip for the whole run, not one per tile), then the whole wavefront goes through the
    // tiles that failed, point by point as before.  After a farthest-first pick almost every tile passes: the sweep of a
    // step was 9 of its 18 us, all of it reading best[] and nearest[].
    auto sweep_tile = [&](int64_t tile) {  // the whole wavefront: four points per lane, and the tile's new summary
        const int64_t g = tile * 64 + lane;
        const bool act = g < n4;
        unsigned long long tkey = ~0ull;
        float tmax = -__int_as_float(0x7f800000);
        unsigned tmask = 0u;
        if (act) {
            const float4 b4 = reinterpret_cast<const float4 *>(best)[g];
            const unsigned l4 = reinterpret_cast<const unsigned *>(nearest)[g];
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
            const unsigned long long before = key;
            key = ~0ull;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                // (a NaN or +inf in b, E or Delta makes the comparison false: full evaluation)
                const float w = E - bb[v];
                const int lab = (int)((l4 >> (8 * v)) & 0xffu);
                const bool skip = w >= 0.0f && sDelta[lab] >= 4.0001f * w;
                float b_after = bb[v];
                int lab_after = lab;
                visit(4 * g + v, bb[v], skip, b_after, lab_after);
                tmax = b_after != b_after ? __int_as_float(0x7f800000) : fmaxf(tmax, b_after);
                tmask |= 1u << (lab_after & 31);
            }
            tkey = key;
            key = tkey < before ? tkey : before;
        }
        if (meta) {
            tkey = wave_min_u64_lane0(tkey);
#define ET_DOWN(O)                                                                          \
    do {                                                                                    \
        tmax = fmaxf(tmax, __uint_as_float(lane_down_u32<O>(__float_as_uint(tmax))));       \
        tmask |= lane_down_u32<O>(tmask);                                                   \
    } while (0)
            ET_DOWN(32);
            ET_DOWN(16);
            ET_DOWN(8);
            ET_DOWN(4);
            ET_DOWN(2);
            ET_DOWN(1);
#undef ET_DOWN
            if (lane == 0)
                meta[tile] = make_uint4((unsigned)(tkey & 0xffffffffull), (unsigned)(tkey >> 32), __float_as_uint(tmax), tmask);
        }
    };
    if (meta && vec) {
        for (int64_t tb = t_begin; tb < t_end; tb += 64) {  // (wave-uniform)
            const int64_t mine = tb + lane;
            bool todo_mine = mine < t_end;
            if (meta_valid && todo_mine) {
                const uint4 m = tb == t_begin ? meta0 : meta[mine];
                const unsigned ob = m.y;  // orderable(b_min) -> b_min
                const float b_min = __uint_as_float((ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob), b_max = __uint_as_float(m.z);
                float dmin = __int_as_float(0x7f800000);
                for (unsigned bits = m.w; bits; bits &= bits - 1u) dmin = fminf(dmin, sDelta[__builtin_ctz(bits)]);
                // (a NaN anywhere makes a comparison false: the tile is looked at point by point)
                if (m.w != 0u && E - b_max >= 0.0f && dmin >= 4.0001f * (E - b_min)) {
                    const unsigned long long k = ((unsigned long long)m.y << 32) | m.x;
                    key = k < key ? k : key;
                    todo_mine = false;
                }
            }
            for (unsigned long long todo = __ballot(todo_mine); todo; todo &= todo - 1ull)
                sweep_tile(tb + __builtin_ctzll(todo));
        }
    } else {
        for (int64_t g = tid; g < n4; g += stride) {
            const float4 b4 = reinterpret_cast<const float4 *>(best)[g];
            const unsigned l4 = reinterpret_cast<const unsigned *>(nearest)[g];
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                // (a NaN or +inf in b, E or Delta makes the comparison false: full evaluation)
                const float w = E - bb[v];
                const bool skip = w >= 0.0f && sDelta[(l4 >> (8 * v)) & 0xffu] >= 4.0001f * w;
                float b_after;
                int lab_after = 0;
                visit(4 * g + v, bb[v], skip, b_after, lab_after);
            }
        }
    }
    for (int64_t n = 4 * n4 + tid; n < N; n += stride) {
        float b = 0.f;
        bool skip = false;
        if (step > 1) {
            b = best[n];
            const float w = E - b;
            skip = w >= 0.0f && sDelta[(int)nearest[n]] >= 4.0001f * w;
        }
        float b_after;
        int lab_after;
        visit(n, b, skip, b_after, lab_after);
    }
    key = wave_min_u64_lane0(key);
    if (step == 1) {  // (the largest |x|: made in the first step only)
        mabs = fmaxf(mabs, __uint_as_float(lane_down_u32<32>(__float_as_uint(mabs))));
        mabs = fmaxf(mabs, __uint_as_float(lane_down_u32<16>(__float_as_uint(mabs))));
        mabs = fmaxf(mabs, __uint_as_float(lane_down_u32<8>(__float_as_uint(mabs))));
        mabs = fmaxf(mabs, __uint_as_float(lane_down_u32<4>(__float_as_uint(mabs))));
        mabs = fmaxf(mabs, __uint_as_float(lane_down_u32<2>(__float_as_uint(mabs))));
        mabs = fmaxf(mabs, __uint_as_float(lane_down_u32<1>(__float_as_uint(mabs))));
    }
    if ((threadIdx.x & 63) == 0) {
        sKey[threadIdx.x >> 6] = key;
        sMax[threadIdx.x >> 6] = __float_as_uint(mabs);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned mb = sMax[0];
        for (int w = 1; w < kKmThreads / 64; ++w) {
            key = sKey[w] < key ? sKey[w] : key;
            mb = sMax[w] > mb ? sMax[w] : mb;
        }
        if (PERSIST) __hip_atomic_store(&block_keys[blockIdx.x], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else block_keys[blockIdx.x] = key;
        // non-negative floats order like their bits; only a workgroup that would raise the maximum touches it
        if (step == 1 && mb > __hip_atomic_load(max_abs_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_abs_bits, mb);
    }
}

template <int D>
__global__ __launch_bounds__(kKmThreads) void kmeans_init_step_kernel(const float *__restrict__ X, int64_t N, int d_rt,
                                                                      int K, int step, const float *__restrict__ C0,
                                                                      float *__restrict__ best,
                                                                      uint8_t *__restrict__ nearest,
                                                                      unsigned *__restrict__ max_abs_bits,
                                                                      int64_t index_base,
                                                                      unsigned long long *__restrict__ block_keys,
                                                                      const unsigned long long *__restrict__ prev_keys,
                                                                      int n_prev, float *C0_rw, unsigned char *cand,
                                                                      uint4 *__restrict__ meta, int meta_valid) {
    init_step_body<D, false>(X, N, d_rt, K, step, C0, best, nearest, max_abs_bits, index_base, block_keys, prev_keys, n_prev,
                             C0_rw, cand, meta, meta_valid);
}

// reduce the workgroup keys; candidate record = {key, d floats of the winning local point}
__global__ __launch_bounds__(kKmThreads) void kmeans_init_pick_kernel(const float *__restrict__ X, int64_t N, int d,
                                                                      const unsigned long long *__restrict__ block_keys,
                                                                      int n_blocks, int64_t index_base,
                                                                      unsigned char *__restrict__ cand, float *C0_out, int K,
                                                                      int col) {
    __shared__ unsigned long long sKey[kKmThreads / 64];
    unsigned long long key = ~0ull;
    for (int b = threadIdx.x; b < n_blocks; b += kKmThreads) key = block_keys[b] < key ? block_keys[b] : key;
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o);
        key = other < key ? other : key;
    }
    if ((threadIdx.x & 63) == 0) sKey[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kKmThreads / 64; ++w) key = sKey[w] < key ? sKey[w] : key;
        *reinterpret_cast<unsigned long long *>(cand) = key;
        float *pt = reinterpret_cast<float *>(cand + 8);
        const int64_t local = (int64_t)(unsigned)(key & 0xffffffffull) - index_base;
        for (int i = 0; i < d; ++i) {
            pt[i] = (key != ~0ull && local >= 0 && local < N) ? X[(int64_t)i * N + local] : __int_as_float(0x7fc00000);
            if (C0_out) C0_out[i * K + col] = pt[i];  // single-GPU path: the candidate IS the new centroid
        }
    }
}

// sharded farthest-first step: the smallest 64-bit key among the ranks' candidate records (value first, then global
// index: the same winner on every rank) becomes centroid `col`.  One wavefront; replaces a handful of tensor ops.
__global__ void kmeans_init_select_kernel(const unsigned char *__restrict__ cands, int n_cands, int stride, int d, int K,
                                          int col, float *__restrict__ C0) {
    const int lane = threadIdx.x;
    unsigned long long key = ~0ull;
    int who = 0;
    for (int r = lane; r < n_cands; r += 64) {
        const unsigned long long k = *reinterpret_cast<const unsigned long long *>(cands + (size_t)r * stride);
        if (k < key) {
            key = k;
            who = r;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ok = __shfl_xor(key, o);
        const int ow = __shfl_xor(who, o);
        if (ok < key || (ok == key && ow < who)) {
            key = ok;
            who = ow;
        }
    }
    if (lane < d) C0[lane * K + col] = *reinterpret_cast<const float *>(cands + (size_t)who * stride + 8 + 4 * lane);
}

__global__ void kmeans_init_set_kernel(float *__restrict__ C0, int d, int K, int col, const float *__restrict__ point) {
    const int i = threadIdx.x;
    if (i < d) C0[i * K + col] = point[i];
}

__global__ void kmeans_gather_point_kernel(const float *__restrict__ X, int64_t N, int d, int64_t idx,
                                           float *__restrict__ point) {
    const int i = threadIdx.x;
    if (i < d) point[i] = X[(int64_t)i * N + idx];
}

// farthest-first, before its first step, in ONE launch: the first centroid = point `idx` (-> column 0 of C0 and the
// candidate record's point slot) and the running max |x| cleared (a gather kernel, a set kernel and a memset were three
// ~5 us packets with a kernel boundary each)
__global__ void kmeans_init_first_kernel(const float *__restrict__ X, int64_t N, int d, int K, int64_t idx,
                                         float *__restrict__ C0, float *__restrict__ point, unsigned *__restrict__ maxabs) {
    const int i = threadIdx.x;
    if (i < d) {
        const float v = X[(int64_t)i * N + idx];
        point[i] = v;
        C0[i * K] = v;
    }
    if (i == 0) *maxabs = 0u;
}

// the state block before a scan: all zero, "no non-zero value yet" = +inf (two memsets were two packets)
__global__ void kmeans_state_reset_kernel(et_kmeans_state *state) {
    constexpr int kWords = (int)(sizeof(et_kmeans_state) / sizeof(unsigned));
    for (int i = threadIdx.x; i < kWords; i += blockDim.x) reinterpret_cast<unsigned *>(state)[i] = 0u;
    __syncthreads();
    if (threadIdx.x == 0) *reinterpret_cast<unsigned *>(&state->min_nz_x_bits) = 0x7f800000u;
}

static int km_grid(int64_t work_items) {
    const int64_t b = ceil_div(work_items, (int64_t)kKmThreads);
    return (int)(b < 1 ? 1 : (b > kKmMaxBlocks ? kKmMaxBlocks : b));
}

// Grid of a grid-stride kernel sized to exactly one resident wave of workgroups (CUs x workgroups
// per CU from the occupancy query): every workgroup then gets the same number of passes (+-1) and
// there is no sparsely filled last round (4096 workgroups at 5 resident per CU would leave the
// chip 80 % idle for its fourth round).
// CU count of the CURRENT device (cached per device id; a process may drive several GPUs)
static int km_cu_count(int *dev_out = nullptr) {
    static int cu_of_device[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev_out) *dev_out = dev;
    int &n_cu = cu_of_device[dev & 63];
    if (n_cu == 0) {
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    }
    return n_cu;
}

template <typename Kernel>
static int km_resident_grid(Kernel kernel, size_t lds_bytes, int64_t work_items, int threads = kKmThreads) {
    const int n_cu = km_cu_count();
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds_bytes) != hipSuccess || per_cu < 1)
        per_cu = threads > 256 ? 1 : 4;
    int64_t g = (int64_t)n_cu * per_cu;
    if (g > kKmMaxBlocks) g = kKmMaxBlocks;
    const int64_t need = ceil_div(work_items, (int64_t)threads);
    return (int)(need < 1 ? 1 : (need < g ? need : g));
}

static int cpitch_host(int d) { return (d + 1 + 3) & ~3; }
static bool km_dims_ok(int d, int K) { return d >= 1 && d <= ET_KMEANS_MAX_D && K >= 1 && K <= ET_KMEANS_MAX_CLUSTERS; }

static size_t km_plen(int d, int K) { return (size_t)d * K + K + 2; }

// workspace carve: [block partials | block keys | cand | best (N) | labels_u8 (N) | partials | C0 scratch ...]
struct KmWorkspace {
    long long *block_partials;
    unsigned long long *block_keys;
    unsigned long long *block_keys2;  // farthest-first, single GPU: the previous step's keys (read by the next step)
    unsigned char *cand;
    long long *partials;
    et_kmeans_state *state;
    float *best;
    uint8_t *labels_u8;
    unsigned *ticket;        // arrival counter of the fused reduce + update kernel
    unsigned *init_maxabs;   // farthest-first: fp32 bits of max|x| of this shard (collected by step 1)
    uint4 *init_meta;        // farthest-first: one 16-byte summary per 256 points (kmeans_init_step_kernel)
    long long *acc_lanes;    // single-GPU fit: kAccLanes copies of every total, the assignment kernel's atomics land here
    float *last;             // single-GPU fit: centroids (d*K floats) + sim_frac (int64) of the last assignment
    long long *sim_total;    // kmeans_inertia_kernel: the exact similarity sum and the non-finite count
    et_kmeans_state *chain_state[2];  // kmeans_lloyd_chain_kernel: two copies of state / centroids / totals,
    float *chain_cen[2];              // three of the 16-copy delta table (see LloydChain)
    long long *chain_tot[2];
    long long *chain_lanes[3];
    unsigned *persist_ctl;   // kmeans_lloyd_persist_kernel: {arrivals, abort flag}, a cache line of their own
    // packed copy of the points for the trace-less chained loop (kmeans_pack_kernel); nullptr when the shape has none
    PackedHeader *pk_hdr;
    unsigned *pk_xh;
    unsigned short *pk_rr;
    float4 *pk_xa;
    size_t bytes;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// shards the packed copy is kept for: the filter's shape (d = 6, 3 <= K <= 32) and enough points.  Same-box A/B of the
// bench step over shard sizes (tools/ab_packed_sizes.sh, 100 iterations): 3e5 1.41 against 1.28 ms with the fp32 filter,
// 5e5 1.60 / 1.46, 1e6 1.85 / 1.70, 2e6 2.19 / 2.19, 4e6 2.77 / 2.90, 1e7 4.40 / 5.05 -- the packed body's longer set-up
// (label table, accumulator copies) costs ~1.4 us per launch, the bytes it saves only count once a launch streams for longer
// (same-box sweep after the pass was rebuilt in round 5, bench step of 100 iterations, packed against fp32 filter: 1.0e6 points
// 2.14 / 2.06 ms, 1.5e6 2.33 / 2.43, 2.0e6 2.51 / 2.64 -- profiles/r05m)
// (... and again at the round's end -- two delta-table copies, the half-wave last drain, no private segment: 1.5e5 points 1.44 /
// 1.53 ms, 2e5 1.45 / 1.54, 4e5 1.55 / 1.62, 1e6 1.86 / 1.95; 7e4 and 1e5 level, 4e4 1.24 / 1.26 -- profiles/r05m item 14)
constexpr int64_t kPackedMinPoints = 131072;  // 2^17
static int64_t km_packed_min_points() {  // option kmeans_packed_min: tests run the packed path on small shards
    const int64_t v = options().kmeans_packed_min.load(std::memory_order_relaxed);
    return v >= 1024 ? v : kPackedMinPoints;
}
static bool km_packed_shape(int64_t N, int d, int K) { return d == 6 && K >= 3 && K <= 32 && N >= km_packed_min_points() && N % 4 == 0; }

static KmWorkspace km_carve(void *base, int64_t N, int d, int K) {
    KmWorkspace w;
    size_t off = 0;
    unsigned char *p = (unsigned char *)base;
    w.block_partials = (long long *)(p + off);
    off = align_up(off + sizeof(long long) * km_plen(d, K) * kKmMaxBlocks, 256);
    w.block_keys = (unsigned long long *)(p + off);
    off = align_up(off + sizeof(unsigned long long) * kKmMaxBlocks, 256);
    w.block_keys2 = (unsigned long long *)(p + off);
    off = align_up(off + sizeof(unsigned long long) * kKmMaxBlocks, 256);
    w.cand = p + off;
    off = align_up(off + 8 + sizeof(float) * ET_KMEANS_MAX_D, 256);
    w.partials = (long long *)(p + off);
    off = align_up(off + sizeof(long long) * km_plen(d, K), 256);
    w.state = (et_kmeans_state *)(p + off);
    off = align_up(off + sizeof(et_kmeans_state), 256);
    w.best = (float *)(p + off);
    off = align_up(off + sizeof(float) * (size_t)(N > 0 ? N : 1), 256);
    w.labels_u8 = (uint8_t *)(p + off);
    off = align_up(off + (size_t)(N > 0 ? N : 1) + 4, 256);
    w.ticket = (unsigned *)(p + off);
    off = align_up(off + sizeof(unsigned), 256);
    w.init_maxabs = (unsigned *)(p + off);
    off = align_up(off + sizeof(unsigned), 256);
    w.init_meta = (uint4 *)(p + off);
    off = align_up(off + sizeof(uint4) * (size_t)((N > 0 ? N : 1) / 256 + 2), 256);
    w.acc_lanes = (long long *)(p + off);
    off = align_up(off + sizeof(long long) * km_plen(d, K) * 16, 256);
    w.last = (float *)(p + off);
    off = align_up(off + sizeof(float) * (((size_t)d * K + 1) & ~(size_t)1) + sizeof(long long), 256);
    w.sim_total = (long long *)(p + off);
    off = align_up(off + 3 * sizeof(long long), 256);  // (sum, non-finite count, "the last launch made the sum")
    for (int i = 0; i < 2; ++i) {
        w.chain_state[i] = (et_kmeans_state *)(p + off);
        off = align_up(off + sizeof(et_kmeans_state), 256);
        w.chain_cen[i] = (float *)(p + off);
        off = align_up(off + sizeof(float) * (size_t)d * K, 256);
        w.chain_tot[i] = (long long *)(p + off);
        off = align_up(off + sizeof(long long) * km_plen(d, K), 256);
    }
    for (int i = 0; i < 3; ++i) {
        w.chain_lanes[i] = (long long *)(p + off);
        off = align_up(off + sizeof(long long) * km_plen(d, K) * 16, 256);
    }
    w.persist_ctl = (unsigned *)(p + off);
    off = align_up(off + 2 * sizeof(unsigned), 256);
    w.pk_hdr = nullptr;
    w.pk_xh = nullptr;
    w.pk_rr = nullptr;
    w.pk_xa = nullptr;
    if (km_packed_shape(N, d, K)) {  // 46 B per point
        w.pk_hdr = (PackedHeader *)(p + off);
        off = align_up(off + sizeof(PackedHeader), 256);
        w.pk_xh = (unsigned *)(p + off);
        off = align_up(off + 12 * (size_t)N, 256);
        w.pk_rr = (unsigned short *)(p + off);
        off = align_up(off + 2 * (size_t)N, 256);
        w.pk_xa = (float4 *)(p + off);
        off = align_up(off + 8192 * (((size_t)N + 255) / 256), 256);  // (whole blocks of 256 points: xa_index)
    }
    w.bytes = off;
    return w;
}

template <int D>
static int launch_assign(const float *X, int64_t N, int d, int K, const et_kmeans_state *state, const float *cen,
                         const int64_t *given, uint8_t *labels, long long *block_partials, bool vec4, hipStream_t st) {
    const size_t plen = km_plen(d, K);
    const size_t lds = sizeof(long long) * ((plen + 1) & ~(size_t)1) + sizeof(float) * (size_t)K * ((d + 1 + 3) & ~3);
    int grid;
    if (vec4) {
        grid = km_resident_grid(kmeans_assign_kernel<D, 4>, lds, N / 4);
        hipLaunchKernelGGL((kmeans_assign_kernel<D, 4>), dim3(grid), dim3(kKmThreads), lds, st, X, N, d, K, state, cen,
                           given, labels, block_partials);
    } else {
        grid = km_resident_grid(kmeans_assign_kernel<D, 1>, lds, N);
        hipLaunchKernelGGL((kmeans_assign_kernel<D, 1>), dim3(grid), dim3(kKmThreads), lds, st, X, N, d, K, state, cen,
                           given, labels, block_partials);
    }
    return grid;
}

}  // namespace et

using namespace et;

extern "C" size_t et_kmeans_partials_len(int d, int K) { return km_plen(d, K); }

extern "C" size_t et_kmeans_workspace_bytes(int64_t N, int d, int K) {
    if (!km_dims_ok(d, K) || N < 0) return 0;
    return km_carve(nullptr, N, d, K).bytes;
}

extern "C" int et_kmeans_scan(const float *X, int64_t N, int d, et_kmeans_state *state, et_stream_t stream) {
    if (!state || N < 0 || d < 1 || d > ET_KMEANS_MAX_D || (N > 0 && !X)) return ET_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(kmeans_state_reset_kernel, dim3(1), dim3(64), 0, st, state);  // zero; min_nz_x_bits = "+inf"
    ET_LAUNCH_CHECK();
    if (N == 0) return ET_OK;
    const int64_t scan_blocks = ceil_div(N * d / 4 + 1, (int64_t)kKmThreads);
    hipLaunchKernelGGL(kmeans_scan_kernel, dim3((unsigned)(scan_blocks < 1024 ? scan_blocks : 1024)), dim3(kKmThreads), 0, st,
                       X, N * d, state);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_kmeans_begin(et_kmeans_state *state, int64_t n_total, const float *centroids, int d, int K,
                               et_stream_t stream) {
    if (!state || !centroids || n_total < 0 || !km_dims_ok(d, K)) return ET_ERR_INVALID_ARG;
    hipLaunchKernelGGL(kmeans_begin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, n_total, centroids, d, K);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

// the switches of et_options.h (et_set_option), read per fit: same-process A/B runs and tests
static char km_argmax_mode() { return (char)options().kmeans_argmax.load(std::memory_order_relaxed); }
static bool km_packed_mode() { return options().kmeans_packed.load(std::memory_order_relaxed) != 0; }
static bool km_init_tiles_mode() { return options().kmeans_init_tiles.load(std::memory_order_relaxed) != 0; }
static bool km_pack_fused_mode() { return options().kmeans_pack_fused.load(std::memory_order_relaxed) != 0; }
static std::atomic<long long> g_packed_fits{0};  // fits that iterated on the packed copy (tests: the path under test ran)
#ifdef ET_TEST_HOOKS  // libetamd_testhooks.so only: problems of et_kmeans_fit_batch to treat as timed out (bit mask)
static std::atomic<unsigned long long> g_test_abort_mask{0};
extern "C" void et_testhook_kmeans_abort_mask(unsigned long long mask) { g_test_abort_mask.store(mask, std::memory_order_relaxed); }
#endif

// matrix-core filter + exact certification (default; option kmeans_argmax = v disables it)
static bool km_use_filter(const float *X, int64_t N, int d, int K, const uint8_t *labels_u8) {
    const bool vec4 = (N % 4 == 0) && aligned16(X) && ((reinterpret_cast<uintptr_t>(labels_u8) & 3u) == 0);
    return km_argmax_mode() == 'f' && vec4 && d == 6 && K >= 3 && K <= 32 && N >= 1024 && N <= 0xffffffffll;
}

// 96 KB of dynamic LDS for the fat (16-wavefront) kernels: above the default 64 KB window; the attribute is per device
static int km_fat_lds_attribute() {
    static bool lds_set[64] = {};
    int dev_id = 0;
    ET_HIP_TRY(hipGetDevice(&dev_id));
    bool &lds_ok = lds_set[dev_id & 63];
    if (lds_ok) return ET_OK;
#define ET_FAT4(KERNEL)                                                                                       \
    reinterpret_cast<const void *>(KERNEL<10, true>), reinterpret_cast<const void *>(KERNEL<10, false>),           \
        reinterpret_cast<const void *>(KERNEL<16, true>), reinterpret_cast<const void *>(KERNEL<16, false>)
    const void *fat[] = {reinterpret_cast<const void *>(kmeans_assign_filter_kernel<10>),
                         reinterpret_cast<const void *>(kmeans_assign_filter_kernel<16>),
                         ET_FAT4(kmeans_lloyd_chain_kernel), ET_FAT4(kmeans_lloyd_persist_kernel),
                         reinterpret_cast<const void *>(kmeans_lloyd_chain_kernel<10, true, true>),
                         reinterpret_cast<const void *>(kmeans_lloyd_chain_kernel<10, false, true>),
                         reinterpret_cast<const void *>(kmeans_lloyd_chain_kernel<16, true, true>),
                         reinterpret_cast<const void *>(kmeans_lloyd_chain_kernel<16, false, true>)};
#undef ET_FAT4
    for (const void *f : fat) ET_HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    lds_ok = true;
    return ET_OK;
}

static size_t km_filter_lds_bytes(int d, int K, int threads) {
    const size_t plen_ = km_plen(d, K);
    return sizeof(long long) * ((plen_ + 1) & ~(size_t)1) + sizeof(float) * (size_t)K * 8 +
           sizeof(unsigned) * kFilterQueue * (size_t)(threads / 64);
}

// 12 or 16 wavefronts per CU for a shard of N points (one workgroup per CU, 256 points per wavefront pass): the
// launch ends with its slowest wavefront, a pass costs 0.73x as much with three wavefronts per SIMD as with four.
static int km_filter_threads(int64_t N) {
    {  // option kmeans_filter_threads: measurement aid (tools/ab_threads_sizes.sh)
        const int t = options().kmeans_filter_threads.load(std::memory_order_relaxed);
        if (t >= 256 && t <= kFilterMaxThreads && t % 64 == 0) return t;
    }
    int dev = 0, n_cu = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        n_cu <= 0)
        n_cu = 256;
    const int64_t groups = ceil_div(N, (int64_t)256);
    const int64_t p12 = ceil_div(groups, (int64_t)n_cu * (kFilterMinThreads / 64));
    const int64_t p16 = ceil_div(groups, (int64_t)n_cu * (kFilterMaxThreads / 64));
    return (double)p12 * 0.73 < (double)p16 ? kFilterMinThreads : kFilterMaxThreads;
}

// The loops of a single-GPU fit (chained / persistent kernel) on small shards: a launch of 256 x 12 wavefronts for a
// shard that has one or two 256-point passes per CU spends its time placing wavefronts.  Same-box sweep of the bench step
// (tools/ab_threads_sizes.sh, 100 Lloyd iterations, ms): chained 256 / 512 / 768 threads per workgroup at N = 2e4
// 1.01 / 1.10 / 1.19, 7e4 1.07 / 1.14 / 1.23, 1e5 1.09 / 1.16 / 1.24, 2e5 1.25 / 1.20 / 1.29, 3e5 1.37 / 1.24 / 1.27,
// 1e6 2.65 / 1.97 / 1.65 (1024: 1.69); persistent at 2e4 0.96 / 0.99 / 1.07, at 7e4 1.50 / 1.11 / 1.17.
// -> the fewest wavefronts that still give every 256-point pass a wavefront of its own twice over.
static int km_loop_threads(int64_t N, bool persistent) {
    {  // option kmeans_filter_threads: measurement aid
        const int t = options().kmeans_filter_threads.load(std::memory_order_relaxed);
        if (t >= 256 && t <= kFilterMaxThreads && t % 64 == 0) return t;
    }
    const int n_cu = km_cu_count();
    const int64_t groups = ceil_div(N, (int64_t)256);
    if (persistent) return N <= 32768 ? 256 : 512;
    for (int t = 256; t < kFilterMinThreads; t += 256)
        if ((int64_t)n_cu * (t / 64) >= 2 * groups) return t;
    return kFilterMinThreads;
}

static int assign_accumulate_impl(const float *X, int64_t N, int d, int K, et_kmeans_state *state,
                                  const float *centroids, const int64_t *given_labels, uint8_t *labels_u8,
                                  int64_t *partials, void *workspace, size_t workspace_bytes, hipStream_t st,
                                  hipEvent_t ev_begin, hipEvent_t ev_end, bool fused_update = false, float tol = 0.f,
                                  float *trace = nullptr, bool want_sim = true) {
    if (!km_dims_ok(d, K) || N < 0 || !state || !centroids || !partials || (N > 0 && (!X || !labels_u8)))
        return ET_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < et_kmeans_workspace_bytes(N, d, K)) return ET_ERR_WORKSPACE;
    const KmWorkspace w = km_carve(workspace, N, d, K);
    const bool vec4 = (N % 4 == 0) && aligned16(X) && ((reinterpret_cast<uintptr_t>(labels_u8) & 3u) == 0);
    // The filter kernel itself runs the plain exact scan for the first iteration of a fit (state->iter == 0: no
    // labels to confirm yet).
    const bool use_filter = !given_labels && km_use_filter(X, N, d, K, labels_u8);
    int grid = 1;
    if (ev_begin) ET_HIP_TRY(hipEventRecord(ev_begin, st));
    if (use_filter) {
        int rc_attr = km_fat_lds_attribute();
        if (rc_attr) return rc_attr;
        const int threads = km_filter_threads(N);
        const size_t lds = km_filter_lds_bytes(d, K, threads);
        if (K <= 20) {
            grid = km_resident_grid(kmeans_assign_filter_kernel<10>, lds, N / 4, threads);
            hipLaunchKernelGGL(kmeans_assign_filter_kernel<10>, dim3(grid), dim3(threads), lds, st, X, N, K, state,
                               centroids, labels_u8, w.block_partials, (long long *)nullptr);
        } else {
            grid = km_resident_grid(kmeans_assign_filter_kernel<16>, lds, N / 4, threads);
            hipLaunchKernelGGL(kmeans_assign_filter_kernel<16>, dim3(grid), dim3(threads), lds, st, X, N, K, state,
                               centroids, labels_u8, w.block_partials, (long long *)nullptr);
        }
    } else if (N > 0) {
        grid = d == 6 ? launch_assign<6>(X, N, d, K, state, centroids, given_labels, labels_u8, w.block_partials, vec4, st)
                      : launch_assign<0>(X, N, d, K, state, centroids, given_labels, labels_u8, w.block_partials, vec4, st);
    } else {
        grid = launch_assign<0>(X, N, d, K, state, centroids, given_labels, labels_u8, w.block_partials, false, st);
    }
    ET_LAUNCH_CHECK();
    if (ev_end) ET_HIP_TRY(hipEventRecord(ev_end, st));
    const int plen = (int)km_plen(d, K);
    if (fused_update) {
        const size_t lds = sizeof(float) * 2 * (size_t)d * K;
        if (lds > 48 * 1024)
            ET_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kmeans_reduce_update_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kmeans_reduce_update_kernel, dim3((plen + kKmThreads / 64 - 1) / (kKmThreads / 64)),
                           dim3(kKmThreads), lds, st, w.block_partials, grid, plen, state, (long long *)partials, w.ticket, d,
                           K, tol, const_cast<float *>(centroids), trace, w.last);
    } else {
        hipLaunchKernelGGL(kmeans_reduce_partials_kernel, dim3(plen), dim3(kKmThreads), 0, st, w.block_partials, grid,
                           plen, given_labels ? 1 : 0, state, w.partials, (long long *)partials);
    }
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_kmeans_assign_accumulate(const float *X, int64_t N, int d, int K, const et_kmeans_state *state,
                                           const float *centroids, const int64_t *given_labels, uint8_t *labels_u8,
                                           int64_t *partials, void *workspace, size_t workspace_bytes,
                                           et_stream_t stream) {
    return assign_accumulate_impl(X, N, d, K, const_cast<et_kmeans_state *>(state), centroids, given_labels, labels_u8,
                                  partials, workspace, workspace_bytes, (hipStream_t)stream, nullptr, nullptr);
}

extern "C" int et_kmeans_update(et_kmeans_state *state, const int64_t *partials, int d, int K, float tol,
                                float *centroids, float *trace, et_stream_t stream) {
    if (!state || !partials || !centroids || !km_dims_ok(d, K)) return ET_ERR_INVALID_ARG;
    const size_t lds = sizeof(float) * 2 * (size_t)d * K;
    if (lds > 48 * 1024)
        ET_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kmeans_update_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kmeans_update_kernel, dim3(1), dim3(kKmThreads), lds, (hipStream_t)stream, state,
                       (const long long *)partials, d, K, tol, centroids, trace);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_kmeans_joint_done(et_kmeans_state *const *states, int n_problems, float tol, et_stream_t stream) {
    if (!states || n_problems < 1) return ET_ERR_INVALID_ARG;
    hipLaunchKernelGGL(kmeans_joint_done_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, states, n_problems, tol);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_kmeans_labels_i64(const uint8_t *labels_u8, int64_t N, int64_t *labels, et_stream_t stream) {
    if (N < 0 || (N > 0 && (!labels_u8 || !labels))) return ET_ERR_INVALID_ARG;
    if (N == 0) return ET_OK;
    hipLaunchKernelGGL(kmeans_labels_i64_kernel, dim3(km_grid(N / 4 + 1)), dim3(kKmThreads), 0, (hipStream_t)stream,
                       labels_u8, N, labels);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_kmeans_predict_batch(const float *X, int64_t x_stride, int64_t batch, int64_t N, int d,
                                       const float *centroids, int K, int64_t *labels, float *maxsims, et_stream_t stream) {
    if (!km_dims_ok(d, K) || N < 0 || batch < 0 || batch > 65535 || x_stride < 0 || !centroids || (batch * N > 0 && !X))
        return ET_ERR_INVALID_ARG;
    if (batch * N == 0) return ET_OK;
    const size_t lds = sizeof(float) * (size_t)K * ((d + 1 + 3) & ~3);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)km_grid(N), (unsigned)batch);
    if (d == 6)
        hipLaunchKernelGGL((kmeans_predict_kernel<6>), grid, dim3(kKmThreads), lds, st, X, N, d, centroids, K, labels, maxsims,
                           x_stride);
    else
        hipLaunchKernelGGL((kmeans_predict_kernel<0>), grid, dim3(kKmThreads), lds, st, X, N, d, centroids, K, labels, maxsims,
                           x_stride);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_kmeans_predict(const float *X, int64_t N, int d, const float *centroids, int K, int64_t *labels,
                                 float *maxsims, et_stream_t stream) {
    return et_kmeans_predict_batch(X, 0, 1, N, d, centroids, K, labels, maxsims, stream);
}

// fused != nullptr: single-GPU path, the one-workgroup pick launch also stores the candidate as centroid i of `fused`
// (= C0).  (Letting the last of the 4096 step workgroups do the pick was measured 3x SLOWER: 4096 device-scope
// arrivals on one ticket serialise at ~25 ns each.)
static int init_step_grid(int64_t N, int i) {
    return i == 1 ? min(km_grid(N), 1024) : min(km_grid(N / 4 + 1), 1024);  // steps >= 2: four points per lane
}

// fused != nullptr: single-GPU path.  Step i (>= 2) derives centroid i-1 itself from the keys step i-1 left in the other
// key buffer, so no pick launch separates two steps; `last` adds the pick that stores centroid i of the final step.
static int init_step_impl(const float *X, int64_t N, int d, int K, int i, const float *C0, float *best, int64_t index_base,
                          void *cand, void *workspace, size_t workspace_bytes, et_stream_t stream, float *fused,
                          bool last = true) {
    if (!km_dims_ok(d, K) || N < 0 || i < 1 || i >= K || !C0 || !cand || index_base < 0 ||
        index_base + N > 0xffffffffll || (N > 0 && (!X || !best)))
        return ET_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < et_kmeans_workspace_bytes(N, d, K)) return ET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const KmWorkspace w = km_carve(workspace, N, d, K);
    const int grid = init_step_grid(N, i);
    if (i == 1 && !fused) ET_HIP_TRY(hipMemsetAsync(w.init_maxabs, 0, sizeof(unsigned), st));  // (fused: kmeans_init_first_kernel did)
    // key buffers alternate in the fused path (a step reads its predecessor's keys while it writes its own)
    unsigned long long *keys = (fused && (i & 1)) ? w.block_keys2 : w.block_keys;
    const unsigned long long *prev = (fused && i > 1) ? ((i & 1) ? w.block_keys : w.block_keys2) : nullptr;
    const int n_prev = i > 1 ? init_step_grid(N, i - 1) : 0;
    // tile summaries: written by step 2 and every later one, used from step 3 on (K <= 32: the label set is a 32-bit mask)
    uint4 *meta = (K <= 32 && km_init_tiles_mode()) ? w.init_meta : nullptr;
    if (d == 6)
        hipLaunchKernelGGL((kmeans_init_step_kernel<6>), dim3(grid), dim3(kKmThreads), 0, st, X, N, d, K, i, C0, best,
                           w.labels_u8, w.init_maxabs, index_base, keys, prev, n_prev, fused, (unsigned char *)cand, meta,
                           i > 2 ? 1 : 0);
    else
        hipLaunchKernelGGL((kmeans_init_step_kernel<0>), dim3(grid), dim3(kKmThreads), 0, st, X, N, d, K, i, C0, best,
                           w.labels_u8, w.init_maxabs, index_base, keys, prev, n_prev, fused, (unsigned char *)cand, meta,
                           i > 2 ? 1 : 0);
    ET_LAUNCH_CHECK();
    if (!fused || last) {
        hipLaunchKernelGGL(kmeans_init_pick_kernel, dim3(1), dim3(kKmThreads), 0, st, X, N, d, keys, grid, index_base,
                           (unsigned char *)cand, fused, K, i);
        ET_LAUNCH_CHECK();
    }
    return ET_OK;
}

extern "C" int et_kmeans_init_step(const float *X, int64_t N, int d, int K, int i, const float *C0, float *best,
                                   int64_t index_base, void *cand, void *workspace, size_t workspace_bytes,
                                   et_stream_t stream) {
    return init_step_impl(X, N, d, K, i, C0, best, index_base, cand, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int et_kmeans_init_select(const void *cands, int n_cands, int stride_bytes, int d, int K, int col, float *C0,
                                     et_stream_t stream) {
    if (!cands || !C0 || !km_dims_ok(d, K) || col < 0 || col >= K || n_cands < 1 || stride_bytes < 8 + 4 * d ||
        (stride_bytes & 7) || (reinterpret_cast<uintptr_t>(cands) & 7u))
        return ET_ERR_INVALID_ARG;
    hipLaunchKernelGGL(kmeans_init_select_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                       (const unsigned char *)cands, n_cands, stride_bytes, d, K, col, C0);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_kmeans_init_set(float *C0, int d, int K, int col, const float *point, et_stream_t stream) {
    if (!C0 || !point || !km_dims_ok(d, K) || col < 0 || col >= K) return ET_ERR_INVALID_ARG;
    hipLaunchKernelGGL(kmeans_init_set_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, C0, d, K, col, point);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_kmeans_gather_point(const float *X, int64_t N, int d, int64_t local_index, float *point,
                                      et_stream_t stream) {
    if (!X || !point || d < 1 || d > ET_KMEANS_MAX_D || local_index < 0 || local_index >= N) return ET_ERR_INVALID_ARG;
    hipLaunchKernelGGL(kmeans_gather_point_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, X, N, d, local_index,
                       point);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_kmeans_init_farthest(const float *X, int64_t N, int d, int K, int64_t first_index, float *C0,
                                       void *workspace, size_t workspace_bytes, et_stream_t stream) {
    if (!km_dims_ok(d, K) || N < 1 || !X || !C0 || first_index < 0 || first_index >= N || N > 0xffffffffll)
        return ET_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < et_kmeans_workspace_bytes(N, d, K)) return ET_ERR_WORKSPACE;
    const KmWorkspace w = km_carve(workspace, N, d, K);
    float *pt = reinterpret_cast<float *>(w.cand + 8);
    hipLaunchKernelGGL(kmeans_init_first_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, X, N, d, K, first_index, C0, pt,
                       w.init_maxabs);
    ET_LAUNCH_CHECK();
    int rc = ET_OK;
    // (a one-launch form of steps 2 .. K-1 with a fence-free grid barrier was built in round 4 and lost -- 0.51 against 0.25 ms at
    // 1e7 points, no gain at 1e5: profiles/r04b_init_persist.txt; its source: tools/lost_forms/kmeans_init_persist.hip.txt)
    for (int i = 1; i < K && !rc; ++i)  // one launch per new centroid (+ one pick for the last)
        rc = init_step_impl(X, N, d, K, i, C0, w.best, 0, w.cand, workspace, workspace_bytes, stream, C0, i == K - 1);
    return rc;
}

// Everything the chained loop's buffers need before its first launch, in ONE launch (the six separate copies / fills
// it replaces were ~5 us packets each): state and centroids into copy 0, totals of copy 0 and the three delta tables zeroed.
__global__ __launch_bounds__(kKmThreads) void kmeans_chain_prepare_kernel(const et_kmeans_state *__restrict__ state,
                                                                          const float *__restrict__ cen, int dk, int plen,
                                                                          LloydChain first, long long *lanes_b, long long *lanes_c) {
    const int tid = blockIdx.x * kKmThreads + threadIdx.x, n_thr = gridDim.x * kKmThreads;
    constexpr int kStateWords = (int)(sizeof(et_kmeans_state) / sizeof(unsigned));
    if (tid < kStateWords) reinterpret_cast<unsigned *>(first.st_wr)[tid] = reinterpret_cast<const unsigned *>(state)[tid];
    for (int e = tid; e < dk; e += n_thr) first.cen_wr[e] = cen[e];
    for (int e = tid; e < plen; e += n_thr) first.tot_wr[e] = 0;
    for (int e = tid; e < plen * kAccLanes; e += n_thr) {
        first.lanes_wr[e] = 0;
        lanes_b[e] = 0;
        lanes_c[e] = 0;
    }
}

// A collective the chained loop runs between two launches when the points are sharded over ranks: SUM over ranks of
// `count` int64 values, in place, enqueued on `st` (csrc/et_sharded.hip binds it to ncclAllReduce).  Without one
// (single GPU) the loop also polls the convergence flag opportunistically; with one, every rank must enqueue the same
// collectives, so the flag is read by a blocking wait on a specific, long-arrived copy (et_hostring.h).
struct ChainHook {
    int (*reduce)(void *ctx, long long *buf, size_t count, hipStream_t st) = nullptr;
    void *ctx = nullptr;
};

// The chained Lloyd loop (kmeans_lloyd_chain_kernel): `state` holds the initial state block (after scan / begin) on
// entry and the final one on return, `centroids` the initial / final centroids, `partials` receives the final totals.
// Ends with the update of the last assignment (finalize kernel) and, for trace-less fits, the inertia pass.
static int km_chain_run(const float *X, int64_t N, int d, int K, int max_iter, float tol, float *centroids,
                        uint8_t *labels_u8, float *trace, et_kmeans_state *state, long long *partials, const KmWorkspace &w,
                        hipStream_t st, ChainHook hook, std::vector<hipEvent_t> *events, int time_every, int *launched_out) {
    constexpr int kEvery = 4;
    auto timed = [&](int it) { return events && (it == 0 || it % time_every == 1); };  // (time_every >= kTimedRun)
    int rc = ET_OK;
    StateRing *ring = StateRing::get(&rc);
    if (!ring) return rc;
    const bool want_sim = trace != nullptr;
    const int threads = km_loop_threads(N, false);
    const size_t plen = km_plen(d, K), lds = km_filter_lds_bytes(d, K, threads);
    // rows that allow 16-byte loads and enough points: the filter body; any other shard of a sharded fit: the exact scan,
    // one point per lane, inside the same kernel (a single-GPU fit only comes here with vec_ok)
    const bool vec_ok = km_use_filter(X, N, d, K, labels_u8);
    const int64_t work_items = vec_ok ? N / 4 : N;
    rc = km_fat_lds_attribute();
    if (rc) return rc;
    // single GPU: the kernel reports (done, iterations applied) into the ring's pinned mailbox and nothing is copied
    // inside the loop; sharded: the lockstep state copies below (what a rank reads from a mailbox depends on timing)
    unsigned long long *mail = hook.reduce ? nullptr : ring->mailbox_device();
    if (mail) ring->mailbox_reset();
    constexpr int kAhead = 16;  // launches the host may be ahead of the device's report
    {
        LloydChain first{};  // the kernel writes through the *_wr fields: copy 0 of state / centroids / totals, table 0
        first.st_wr = w.chain_state[0];
        first.cen_wr = w.chain_cen[0];
        first.tot_wr = w.chain_tot[0];
        first.lanes_wr = w.chain_lanes[0];
        hipLaunchKernelGGL(kmeans_chain_prepare_kernel, dim3(8), dim3(kKmThreads), 0, st, (const et_kmeans_state *)state,
                           (const float *)centroids, d * K, (int)plen, first, w.chain_lanes[1], w.chain_lanes[2]);
        ET_LAUNCH_CHECK();
    }
    // trace-less fits of big shards iterate on the packed copy (ET_KMEANS_PACKED=0: the fp32 filter, for A/B runs)
    const bool packed = vec_ok && !want_sim && w.pk_xh && km_packed_mode();
    const bool pack_fused = km_pack_fused_mode();
    if (packed) g_packed_fits.fetch_add(1, std::memory_order_relaxed);
    if (packed && !pack_fused) {
        const int64_t quads = N / 4;
        const int pgrid = (int)std::min<int64_t>((quads + kKmThreads - 1) / kKmThreads, 1024);
        hipLaunchKernelGGL(kmeans_pack_kernel, dim3(pgrid), dim3(kKmThreads), 0, st, X, N, (const et_kmeans_state *)state,
                           w.pk_hdr, w.pk_xh, w.pk_rr, w.pk_xa);
        ET_LAUNCH_CHECK();
    }
    int chain_copies = options().kmeans_chain_copies.load(std::memory_order_relaxed);
    if (chain_copies != 1 && chain_copies != 2 && chain_copies != 4 && chain_copies != 8) chain_copies = 2;
    while (chain_copies > 1 && (size_t)chain_copies * compact_pitch((int)plen) > plen * kAccLanes) chain_copies >>= 1;
    auto chain_for = [&](int t) {
        LloydChain ch;
        ch.st_rd = w.chain_state[t & 1];
        ch.st_wr = w.chain_state[(t + 1) & 1];
        ch.cen_rd = w.chain_cen[t & 1];
        ch.cen_wr = w.chain_cen[(t + 1) & 1];
        ch.tot_rd = w.chain_tot[t & 1];
        ch.tot_wr = w.chain_tot[(t + 1) & 1];
        ch.lanes_rd = w.chain_lanes[t % 3];
        ch.lanes_wr = w.chain_lanes[(t + 1) % 3];
        ch.lanes_zero = w.chain_lanes[(t + 2) % 3];
        ch.last = w.last;
        ch.mail = mail;
        // ONE compact copy of the delta table in every form of the chained loop: measured against the 16-copy table on one
        // GPU it is 1.0 us per launch FASTER (50.0 against 51.0 us, three alternating runs: the prologue reads 142 values
        // instead of folding 2272, and <= 256 arrivals per address spread over the launch's tail are absorbed by the
        // memory side), and it is what a sharded fit puts on the wire
        ch.compact = 1;
        // ... and on one GPU a few of them (option kmeans_chain_copies, default 2): a launch's 256 workgroups add their deltas
        // within a few microseconds of each other, and arrivals on one address are served one after the other
        ch.copies = hook.reduce ? 1 : chain_copies;
        ch.vec_ok = vec_ok ? 1 : 0;
        ch.pk = packed ? LloydPacked{w.pk_xh, w.pk_rr, w.pk_xa, w.pk_hdr, pack_fused ? 1 : 0} : LloydPacked{nullptr, nullptr, nullptr, nullptr, 0};
        return ch;
    };
    int grid = 0, launched = 0;
    bool done = false;
    // A trace-less fit evaluates the inertia of its last assignment in a pass of its own over the points (48 us at 1e7
    // points + two packets).  When the loop runs to max_iter its last launch is known beforehand: that one launch takes the
    // form that accumulates the exact similarity sum (the traced fits' kernel on the fp32 rows, +9 us at 1e7 points), the
    // finalize kernel's update turns the sum into the inertia -- the same integers, the same bits -- and the pass is skipped
    // ON THE DEVICE (sim_total[2]): a fit that converges earlier never reaches that launch's assignment and keeps the pass.
    // (the decision must not depend on this rank's shard: a shard without 16-byte rows runs the exact scan, which adds the
    // similarity sum in every launch -- so every rank of a sharded fit feeds the last launch's sum and skips the pass)
    const bool sim_tail = !want_sim && max_iter >= 2;
    bool last_was_sim = false;
    for (int it = 0; it < max_iter && !done; ++it) {
        const LloydChain ch = chain_for(it);
        const bool sim_now = want_sim || (sim_tail && it == max_iter - 1);
        last_was_sim = sim_now && !want_sim;
        if (timed(it)) ET_HIP_TRY(hipEventRecord((*events)[2 * it], st));
#define ET_LAUNCH_CHAIN(NR, SIM)                                                                                          \
    do {                                                                                                                  \
        if (!grid) {                                                                                                      \
            grid = km_resident_grid(kmeans_lloyd_chain_kernel<NR, SIM>, lds, work_items, threads);                        \
            const int cap = options().kmeans_loop_grid.load(std::memory_order_relaxed);                                   \
            if (cap > 0 && grid > cap) grid = cap;                                                                        \
        }                                                                                                                 \
        if (it == 0)                                                                                                      \
            hipLaunchKernelGGL((kmeans_lloyd_chain_kernel<NR, SIM, true>), dim3(grid), dim3(threads), lds, st, X, N, K,   \
                               ch, labels_u8, tol, trace, 0);                                                             \
        else                                                                                                              \
            hipLaunchKernelGGL((kmeans_lloyd_chain_kernel<NR, SIM, false>), dim3(grid), dim3(threads), lds, st, X, N, K,  \
                               ch, labels_u8, tol, trace, 1);                                                             \
    } while (0)
        if (K <= 20) {
            if (sim_now) ET_LAUNCH_CHAIN(10, true);
            else ET_LAUNCH_CHAIN(10, false);
        } else {
            if (sim_now) ET_LAUNCH_CHAIN(16, true);
            else ET_LAUNCH_CHAIN(16, false);
        }
#undef ET_LAUNCH_CHAIN
        ET_LAUNCH_CHECK();
        // (the end event of a timed launch: right after the first launch -- the exact scan --, after the FOURTH launch of a
        // sampled run of the others: an event between two kernels costs a dispatch gap on each side, and a bracket around
        // one 33-us launch measured 37.5 us where rocprofv3 saw 32.8; four launches per bracket measure the period)
        if (events && it == 0) ET_HIP_TRY(hipEventRecord((*events)[1], st));
        if (events && it >= kTimedRun && timed(it - (kTimedRun - 1)) && it - (kTimedRun - 1) != 0)
            ET_HIP_TRY(hipEventRecord((*events)[2 * (it - (kTimedRun - 1)) + 1], st));
        // sharded: the deltas this launch added onto its (one-copy, compact) table become the sum over all ranks' before
        // the next launch reads them: d K + K + 2 int64, 1.1 KB for d = 6, K = 20
        if (hook.reduce) {
            rc = hook.reduce(hook.ctx, ch.lanes_wr, plen, st);
            if (rc) return rc;
        }
        launched = it + 1;
        if (mail) {
            // launch `it` reports iter = it (it applied the update of assignment it - 1); stay at most kAhead launches
            // ahead of the last report, stop as soon as a report carries the flag
            for (unsigned spins = 0;; ++spins) {
                if (ring->mailbox_done()) {
                    done = true;
                    break;
                }
                if ((long long)launched - ring->mailbox_iter() <= kAhead) break;
                // (a stream query puts a marker into the queue: only as the rare safety net against a lost report --
                // if everything launched so far has finished, what the mailbox says is final)
                if ((spins & 0xfffu) == 0xfffu && hipStreamQuery(st) == hipSuccess) break;
                sched_yield();
            }
        } else {
            if (launched % kEvery == 0) {
                rc = ring->post(ch.st_wr, st, &done);
                if (!rc && hook.reduce && ring->pending() > 1) rc = ring->wait_oldest(&done);  // the same copy on every rank
                if (rc) return rc;
            }
            if (!hook.reduce) ring->poll(&done);
        }
    }
    const LloydChain ch = chain_for(launched);
    const size_t flds = 4096 + sizeof(long long) * plen;
    hipLaunchKernelGGL(kmeans_chain_finalize_kernel, dim3(1), dim3(kKmThreads), flds, st, ch, state, partials, centroids, d,
                       K, tol, trace, launched > 0 ? 1 : 0, want_sim ? (long long *)nullptr : w.sim_total,
                       (last_was_sim && launched == max_iter) ? 1 : 0);
    ET_LAUNCH_CHECK();
    if (!want_sim) {  // inertia of the last assignment (over all ranks' points); the finalize kernel zeroed sim_total
        const size_t ilds = sizeof(float) * (size_t)K * cpitch_host(d);
        // (every workgroup ends with one device-scope atomic on the same address, ~15 ns each: 4096 workgroups made the
        // pass atomic-bound at 64 us; four points per lane and 1024 workgroups stream instead)
        const int igrid = min(km_grid(N / 4 + 1), 1024);
        hipLaunchKernelGGL((kmeans_inertia_kernel<6>), dim3(igrid), dim3(kKmThreads), ilds, st, X, N, d, K,
                           (const float *)w.last, (const uint8_t *)labels_u8, w.sim_total, (int64_t)0, (int64_t)0,
                           (const long long *)(w.sim_total + 2));
        ET_LAUNCH_CHECK();
        if (hook.reduce) {
            rc = hook.reduce(hook.ctx, w.sim_total, 2, st);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(kmeans_inertia_finish_kernel, dim3(1), dim3(64), 0, st, state, (const float *)w.last, d, K,
                           (const long long *)w.sim_total, (int64_t)0, (const long long *)(w.sim_total + 2));
        ET_LAUNCH_CHECK();
    }
    if (launched_out) *launched_out = launched;
    return ET_OK;
}

// ---- the persistent loop (kmeans_lloyd_persist_kernel) ----
// Its grid barrier needs every workgroup of the launch resident at once.  One launch alone always is (the grid is one
// resident round, km_resident_grid); several fits running side by side in this process (the ten initialisations of the
// sklearn recipe, the moving / static clusterings, BatchKMeans' problems -- one host thread and stream each) share the
// CUs through this counter: a fit takes as many CU slots as its grid has workgroups before it launches and gives them
// back after its final synchronisation, so the persistent grids in flight never need more CUs than the device has.
// (Other kernels may occupy CUs for a while -- they end; another PROCESS on the same GPU can break the promise, which
// is what the kernel's time-out and the chained fallback are for.)
#include <condition_variable>
#include <mutex>
namespace et {
class PersistSlots {
  public:
    static PersistSlots &of_device(int dev) {
        static PersistSlots slots[64];
        return slots[dev & 63];
    }
    void acquire(int n, int capacity) {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return used_ == 0 || used_ + n <= capacity; });
        used_ += n;
    }
    void release(int n) {
        {
            std::lock_guard<std::mutex> lk(m_);
            used_ -= n;
        }
        cv_.notify_all();
    }

  private:
    std::mutex m_;
    std::condition_variable cv_;
    int used_ = 0;
};

}  // namespace et
namespace et {

__global__ __launch_bounds__(kKmThreads) void kmeans_persist_prepare_kernel(int plen, long long *l0, long long *l1, long long *l2,
                                                                            unsigned *ctl, long long *sim_total,
                                                                            int64_t ws_stride) {
    {   // blockIdx.y: problem of a batch (workspaces of identical layout, ws_stride bytes apart)
        const int64_t off = (int64_t)blockIdx.y * ws_stride;
        l0 = reinterpret_cast<long long *>(reinterpret_cast<char *>(l0) + off);
        l1 = reinterpret_cast<long long *>(reinterpret_cast<char *>(l1) + off);
        l2 = reinterpret_cast<long long *>(reinterpret_cast<char *>(l2) + off);
        ctl = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(ctl) + off);
        sim_total = reinterpret_cast<long long *>(reinterpret_cast<char *>(sim_total) + off);
    }
    const int tid = blockIdx.x * kKmThreads + threadIdx.x, n_thr = gridDim.x * kKmThreads;
    for (int e = tid; e < plen * kAccLanes; e += n_thr) {
        l0[e] = 0;
        l1[e] = 0;
        l2[e] = 0;
    }
    if (tid < 2) {
        ctl[tid] = 0u;
        sim_total[tid] = 0;
    }
}

// Which loop form a single-GPU fit takes.  Measured per iteration (tools/ab_loop_sizes.py, profiles/r03f_loop_sizes.txt,
// same box): N = 2e4 11.9 us persistent / 13.7 chained, 7e4 12.9 / 13.9, 1e5 13.6 / 14.0, 3e5 16.2 / 13.6,
// 1e6 20.9 / 15.4, 1e7 56.1 / 49.7 -- the grid barrier + fold + update of the persistent form (~4 us for 33 workgroups,
// ~7 us + the spread of 256 workgroups' finishing times for a full grid) beats a kernel boundary only while the grid is
// small; for a full grid the staggered start of a new launch's workgroups happens to hide the uneven pass counts that
// the barrier exposes.  Hence: persistent up to kPersistMaxPoints, chained above; option kmeans_loop = persist / chain forces one.
// (round 3, later: with 256-thread workgroups the chained loop is ahead from ~3e4 points on -- the table above
// km_loop_threads; et_kmeans_fit_batch keeps the persistent form for its side-by-side problems at any size it takes)
constexpr int64_t kPersistMaxPoints = 32768;
static char km_persist_mode() { return (char)options().kmeans_loop.load(std::memory_order_relaxed); }  // 'a'uto, 'c'hain, 'p'ersist
static bool km_persist_wanted(int64_t N) {
    const char mode = km_persist_mode();
    if (mode == 'c') return false;
    if (mode == 'p') return true;
    return N <= kPersistMaxPoints;
}

// All Lloyd iterations of a single-GPU fit in one launch.  Same contract as km_chain_run; *aborted = true (and nothing
// written to state / centroids / partials) when the grid barrier timed out -- the caller repeats the fit chained.
static int km_persist_run(const float *X, int64_t N, int d, int K, int max_iter, float tol, float *centroids,
                          uint8_t *labels_u8, float *trace, et_kmeans_state *state, long long *partials, const KmWorkspace &w,
                          hipStream_t st, hipEvent_t ev_begin, hipEvent_t ev_end, bool *aborted, int *grid_out) {
    const bool want_sim = trace != nullptr;
    const int threads = km_loop_threads(N, true);
    const size_t plen = km_plen(d, K), lds = km_filter_lds_bytes(d, K, threads);
    int rc = km_fat_lds_attribute();
    if (rc) return rc;
    hipLaunchKernelGGL(kmeans_persist_prepare_kernel, dim3(8), dim3(kKmThreads), 0, st, (int)plen, w.chain_lanes[0],
                       w.chain_lanes[1], w.chain_lanes[2], w.persist_ctl, w.sim_total, (int64_t)0);
    ET_LAUNCH_CHECK();
    LloydPersist pa;
    pa.st_in = state;
    pa.cen_in = centroids;
    pa.st_out = w.chain_state[0];  // staged: the caller's buffers are only written once the loop is known to have run
    pa.cen_out = w.chain_cen[0];
    pa.tot_out = w.chain_tot[0];
    pa.lanes0 = w.chain_lanes[0];
    pa.lanes1 = w.chain_lanes[1];
    pa.lanes2 = w.chain_lanes[2];
    pa.arrive = w.persist_ctl;
    pa.abort = w.persist_ctl + 1;
    pa.last = w.last;
    pa.ws_stride = pa.x_stride = pa.cen_stride = 0;
    int grid = 0, dev = 0;
    const int n_cu = km_cu_count(&dev);
    // one resident round of workgroups of the instantiation that is launched
    if (K <= 20) grid = want_sim ? km_resident_grid(kmeans_lloyd_persist_kernel<10, true>, lds, N / 4, threads)
                                 : km_resident_grid(kmeans_lloyd_persist_kernel<10, false>, lds, N / 4, threads);
    else grid = want_sim ? km_resident_grid(kmeans_lloyd_persist_kernel<16, true>, lds, N / 4, threads)
                         : km_resident_grid(kmeans_lloyd_persist_kernel<16, false>, lds, N / 4, threads);
    if (grid > n_cu) grid = n_cu;  // one fat workgroup per CU is what the co-residency accounting assumes
    PersistSlots &slots = PersistSlots::of_device(dev);
    slots.acquire(grid, n_cu);
    struct Release {
        PersistSlots &s;
        int n;
        ~Release() { s.release(n); }
    } release_on_exit{slots, grid};
    if (ev_begin) ET_HIP_TRY(hipEventRecord(ev_begin, st));
#define ET_LAUNCH_PERSIST(NR, SIM)                                                                                \
    hipLaunchKernelGGL((kmeans_lloyd_persist_kernel<NR, SIM>), dim3(grid), dim3(threads), lds, st, X, N, K, pa,      \
                       labels_u8, tol, trace, max_iter)
    if (K <= 20) {
        if (want_sim) ET_LAUNCH_PERSIST(10, true);
        else ET_LAUNCH_PERSIST(10, false);
    } else {
        if (want_sim) ET_LAUNCH_PERSIST(16, true);
        else ET_LAUNCH_PERSIST(16, false);
    }
#undef ET_LAUNCH_PERSIST
    ET_LAUNCH_CHECK();
    if (ev_end) ET_HIP_TRY(hipEventRecord(ev_end, st));
    if (grid_out) *grid_out = grid;
    // the one host round trip of the fit that the chained loop does not have: did the barrier hold?  (pinned staging
    // would save nothing here: the caller synchronises right after this anyway)
    unsigned ctl[2] = {0u, 0u};
    ET_HIP_TRY(hipMemcpyAsync(ctl, w.persist_ctl, sizeof ctl, hipMemcpyDeviceToHost, st));
    ET_HIP_TRY(hipStreamSynchronize(st));
    *aborted = ctl[1] != 0u;
    if (*aborted) return ET_OK;
    ET_HIP_TRY(hipMemcpyAsync(state, w.chain_state[0], sizeof(et_kmeans_state), hipMemcpyDeviceToDevice, st));
    ET_HIP_TRY(hipMemcpyAsync(centroids, w.chain_cen[0], sizeof(float) * (size_t)d * K, hipMemcpyDeviceToDevice, st));
    ET_HIP_TRY(hipMemcpyAsync(partials, w.chain_tot[0], sizeof(long long) * plen, hipMemcpyDeviceToDevice, st));
    if (!want_sim) {  // inertia of the last assignment; the prepare kernel zeroed sim_total
        const size_t ilds = sizeof(float) * (size_t)K * cpitch_host(d);
        const int igrid = min(km_grid(N / 4 + 1), 1024);
        hipLaunchKernelGGL((kmeans_inertia_kernel<6>), dim3(igrid), dim3(kKmThreads), ilds, st, X, N, d, K,
                           (const float *)w.last, (const uint8_t *)labels_u8, w.sim_total);
        hipLaunchKernelGGL(kmeans_inertia_finish_kernel, dim3(1), dim3(64), 0, st, state, (const float *)w.last, d, K,
                           (const long long *)w.sim_total);
        ET_LAUNCH_CHECK();
    }
    return ET_OK;
}
}  // namespace et

#ifdef ET_EXP_WAITSTAMP
extern "C" int et_debug_waitstamp(unsigned long long *host, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(et::g_waitstamp), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(et::g_waitstamp), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
extern "C" int et_debug_prostamp(unsigned long long *host, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(et::g_prostamp), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(et::g_prostamp), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
#endif
#ifdef ET_PERSIST_STAMPS
extern "C" int et_debug_persist_stamps(void *host, size_t bytes) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(et::g_persist_stamps), bytes) == hipSuccess ? 0 : 1;
}
#endif

// Entry points for csrc/et_sharded.hip (not part of the public header): can this rank's shard run the chained loop,
// and the loop itself with a reduction between the launches.  `workspace` as for et_kmeans_fit.
// (the choice depends on d, K and the process-wide ET_KMEANS_ARGMAX setting only -- never on a rank's own shard --, so the
// ranks of a sharded fit agree on the loop form, i.e. on the collectives they enqueue, without exchanging anything)
extern "C" long long et_internal_kmeans_packed_fits(void) { return g_packed_fits.load(std::memory_order_relaxed); }

extern "C" int et_internal_kmeans_chain_usable(int d, int K) {
    return km_dims_ok(d, K) && km_argmax_mode() == 'f' && d == 6 && K >= 3 && K <= 32 ? 1 : 0;
}
extern "C" int et_internal_kmeans_chain_run(const float *X, int64_t N, int d, int K, int max_iter, float tol, float *centroids,
                                            uint8_t *labels_u8, float *trace, et_kmeans_state *state, int64_t *partials,
                                            void *workspace, size_t workspace_bytes,
                                            int (*reduce)(void *, long long *, size_t, hipStream_t), void *ctx,
                                            et_stream_t stream) {
    if (!workspace || workspace_bytes < et_kmeans_workspace_bytes(N, d, K)) return ET_ERR_WORKSPACE;
    const KmWorkspace w = km_carve(workspace, N, d, K);
    ChainHook hook;
    hook.reduce = reduce;
    hook.ctx = ctx;
    return km_chain_run(X, N, d, K, max_iter, tol, centroids, labels_u8, trace, state, (long long *)partials, w,
                        (hipStream_t)stream, hook, nullptr, 8, nullptr);
}

// ---- several fits side by side in ONE persistent launch (blockIdx.y = problem) ----
// The reference's anchor clustering is ten independent small fits (sklearn's n_init = 10, anchor.py:65-71) on the same
// points; at dataset sizes each is latency bound and driving them from ten host threads / streams scaled to barely 2x
// (profiles/r03g_concurrent_inits.txt: 3.4 ms of stream / thread set-up, fits slowed down by each other).  Here the
// problems are the y dimension of ONE persistent grid: each has its own workspace (same layout, ws_stride apart), its
// own barrier counter and stops on its own error; problems whose workgroups do not all fit on the device at once are
// launched in chunks.
namespace et {
__global__ __launch_bounds__(kKmThreads) void kmeans_batch_collect_kernel(const float *cen_staged, const et_kmeans_state *st_staged,
                                                                          et_kmeans_state *st_final, int64_t ws_stride,
                                                                          float *centroids, int dk, const unsigned *ctl) {
    const int64_t off = (int64_t)blockIdx.x * ws_stride;
    // a problem whose grid barrier timed out never wrote its staged results: leave the caller's (initial) centroids and
    // the begun state alone -- the host repeats that fit from them with the chained loop
    if (byte_shift(ctl, off)[1] != 0u) return;
    const float *src = byte_shift(cen_staged, off);
    for (int e = threadIdx.x; e < dk; e += kKmThreads) centroids[(int64_t)blockIdx.x * dk + e] = src[e];
    constexpr int kStateWords = (int)(sizeof(et_kmeans_state) / sizeof(unsigned));
    if ((int)threadIdx.x < kStateWords)
        reinterpret_cast<unsigned *>(byte_shift(st_final, off))[threadIdx.x] =
            reinterpret_cast<const unsigned *>(byte_shift(st_staged, off))[threadIdx.x];
}
}  // namespace et

extern "C" int et_kmeans_fit(const float *X, int64_t N, int d, int K, int max_iter, float tol, float *centroids,
                             int64_t *labels, float *trace, et_kmeans_state *state_host,
                             et_kmeans_timing *timing_host, void *workspace, size_t workspace_bytes,
                             et_stream_t stream);

extern "C" size_t et_kmeans_batch_workspace_bytes(int64_t N, int d, int K, int64_t batch) {
    const size_t one = et_kmeans_workspace_bytes(N, d, K);
    return one == 0 || batch < 1 ? 0 : one * (size_t)batch;
}

extern "C" int et_kmeans_fit_batch(const float *X, int64_t x_stride, int64_t N, int d, int K, int64_t batch, int max_iter,
                                   float tol, float *centroids, int64_t *labels, et_kmeans_state *states_host,
                                   void *workspace, size_t workspace_bytes, et_stream_t stream) {
    if (!km_dims_ok(d, K) || N < 1 || !X || !centroids || !states_host || max_iter < 1 || batch < 1 || batch > 65535 ||
        x_stride < 0)
        return ET_ERR_INVALID_ARG;
    const size_t one = et_kmeans_workspace_bytes(N, d, K);
    if (!workspace || workspace_bytes < one * (size_t)batch) return ET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const KmWorkspace w = km_carve(workspace, N, d, K);  // problem 0's; problem b's is the same layout, b * one bytes on
    const int64_t dk = (int64_t)d * K;
    // every problem alone through et_kmeans_fit: shapes the persistent kernel does not take, or when it is switched off
    // (NaN / Inf in one problem does not stop the others, but is reported: ET_ERR_BAD_DATA after the loop, like the
    // side-by-side path)
    auto one_by_one = [&](int64_t from, int64_t to) -> int {
        bool bad = false;
        for (int64_t b = from; b < to; ++b) {
            const int rc = et_kmeans_fit(X + b * x_stride, N, d, K, max_iter, tol, centroids + b * dk,
                                         labels ? labels + b * N : nullptr, nullptr,
                                         &states_host[b], nullptr, byte_shift((char *)workspace, b * (int64_t)one), one, stream);
            if (rc && rc != ET_ERR_BAD_DATA) return rc;
            bad = bad || rc == ET_ERR_BAD_DATA;
        }
        return bad ? ET_ERR_BAD_DATA : ET_OK;
    };
    bool takes = km_persist_mode() != 'c' && x_stride % 4 == 0;
    for (int64_t b = 0; takes && b < batch; ++b) takes = km_use_filter(X + b * x_stride, N, d, K, w.labels_u8);
    const int threads = km_filter_threads(N);
    const size_t plen = km_plen(d, K), lds = km_filter_lds_bytes(d, K, threads);
    int dev = 0;
    const int n_cu = km_cu_count(&dev);
    int grid = 0;
    if (takes) {
        int rc = km_fat_lds_attribute();
        if (rc) return rc;
        grid = K <= 20 ? km_resident_grid(kmeans_lloyd_persist_kernel<10, false>, lds, N / 4, threads)
                       : km_resident_grid(kmeans_lloyd_persist_kernel<16, false>, lds, N / 4, threads);
        if (grid > n_cu / 2) takes = false;  // a shard that fills the device by itself: nothing to put side by side
    }
    if (!takes) {
        return one_by_one(0, batch);
    }
    const bool shared = x_stride == 0;
    // scale scan: once when the problems share their points, else per problem; then every problem's begin in one launch
    for (int64_t b = 0; b < (shared ? 1 : batch); ++b) {
        const int rc = et_kmeans_scan(X + b * x_stride, N, d, byte_shift(w.state, b * (int64_t)one), stream);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(kmeans_begin_kernel, dim3((unsigned)batch), dim3(64), 0, st, w.state, N, (const float *)centroids, d, K,
                       (int64_t)one, dk, shared ? 1 : 0);
    ET_LAUNCH_CHECK();
    const int per_launch = n_cu / grid;  // problems whose workgroups are resident together (one fat workgroup per CU)
    PersistSlots &slots = PersistSlots::of_device(dev);
    std::vector<unsigned> ctl((size_t)batch * 2, 0u);
    for (int64_t b0 = 0; b0 < batch; b0 += per_launch) {
        const int64_t nb = batch - b0 < per_launch ? batch - b0 : per_launch;
        const int64_t off = b0 * (int64_t)one;
        hipLaunchKernelGGL(kmeans_persist_prepare_kernel, dim3(8, (unsigned)nb), dim3(kKmThreads), 0, st, (int)plen,
                           byte_shift(w.chain_lanes[0], off), byte_shift(w.chain_lanes[1], off),
                           byte_shift(w.chain_lanes[2], off), byte_shift(w.persist_ctl, off), byte_shift(w.sim_total, off),
                           (int64_t)one);
        ET_LAUNCH_CHECK();
        LloydPersist pa;
        pa.st_in = byte_shift(w.state, off);
        pa.cen_in = centroids + b0 * dk;
        pa.st_out = byte_shift(w.chain_state[0], off);
        pa.cen_out = byte_shift(w.chain_cen[0], off);
        pa.tot_out = byte_shift(w.chain_tot[0], off);
        pa.lanes0 = byte_shift(w.chain_lanes[0], off);
        pa.lanes1 = byte_shift(w.chain_lanes[1], off);
        pa.lanes2 = byte_shift(w.chain_lanes[2], off);
        pa.arrive = byte_shift(w.persist_ctl, off);
        pa.abort = byte_shift(w.persist_ctl, off) + 1;
        pa.last = byte_shift(w.last, off);
        pa.ws_stride = (int64_t)one;
        pa.x_stride = x_stride;
        pa.cen_stride = dk;
        slots.acquire(grid * (int)nb, n_cu);
        struct Release {
            PersistSlots &s;
            int n;
            ~Release() { s.release(n); }
        } release_on_exit{slots, grid * (int)nb};
        const dim3 g((unsigned)grid, (unsigned)nb);
        if (K <= 20)
            hipLaunchKernelGGL((kmeans_lloyd_persist_kernel<10, false>), g, dim3(threads), lds, st, X + b0 * x_stride, N, K, pa,
                               byte_shift(w.labels_u8, off), tol, (float *)nullptr, max_iter);
        else
            hipLaunchKernelGGL((kmeans_lloyd_persist_kernel<16, false>), g, dim3(threads), lds, st, X + b0 * x_stride, N, K, pa,
                               byte_shift(w.labels_u8, off), tol, (float *)nullptr, max_iter);
        ET_LAUNCH_CHECK();
#ifdef ET_TEST_HOOKS  // libetamd_testhooks.so only (tests/test_gpu_parity.py): bit mask of problems to treat as timed out
        if (const unsigned long long mask = g_test_abort_mask.load(std::memory_order_relaxed)) {
            for (int64_t b = b0; b < b0 + nb; ++b)
                if (b < 64 && (mask >> b & 1ull))
                    ET_HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(byte_shift(w.persist_ctl, b * (int64_t)one) + 1), 1, 1, st));
        }
#endif
        // did every problem's barrier hold?  (the slots go back when this chunk has run)
        ET_HIP_TRY(hipMemcpy2DAsync(ctl.data() + 2 * b0, 2 * sizeof(unsigned), byte_shift(w.persist_ctl, off), one,
                                    2 * sizeof(unsigned), (size_t)nb, hipMemcpyDeviceToHost, st));
        ET_HIP_TRY(hipStreamSynchronize(st));
    }
    // results: staged centroids / state -> the caller's (B, d, K) array and the problems' state blocks; the inertia of the
    // last assignment; the labels when asked for
    hipLaunchKernelGGL(kmeans_batch_collect_kernel, dim3((unsigned)batch), dim3(kKmThreads), 0, st, (const float *)w.chain_cen[0],
                       (const et_kmeans_state *)w.chain_state[0], w.state, (int64_t)one, centroids, (int)dk,
                       (const unsigned *)w.persist_ctl);
    {
        const size_t ilds = sizeof(float) * (size_t)K * cpitch_host(d);
        const int igrid = min(km_grid(N / 4 + 1), 1024);
        hipLaunchKernelGGL((kmeans_inertia_kernel<6>), dim3(igrid, (unsigned)batch), dim3(kKmThreads), ilds, st, X, N, d, K,
                           (const float *)w.last, (const uint8_t *)w.labels_u8, w.sim_total, (int64_t)one, x_stride);
        hipLaunchKernelGGL(kmeans_inertia_finish_kernel, dim3((unsigned)batch), dim3(64), 0, st, w.state, (const float *)w.last, d,
                           K, (const long long *)w.sim_total, (int64_t)one);
    }
    if (labels)
        hipLaunchKernelGGL(kmeans_labels_i64_kernel, dim3(km_grid(N / 4 + 1), (unsigned)batch), dim3(kKmThreads), 0, st,
                           (const uint8_t *)w.labels_u8, N, labels, (int64_t)one);
    ET_LAUNCH_CHECK();
    ET_HIP_TRY(hipMemcpy2DAsync(states_host, sizeof(et_kmeans_state), w.state, one, sizeof(et_kmeans_state), (size_t)batch,
                                hipMemcpyDeviceToHost, st));
    ET_HIP_TRY(hipStreamSynchronize(st));
    // a problem whose grid barrier timed out (another process on the GPU): that fit again, alone, with the chained loop
    for (int64_t b = 0; b < batch; ++b) {
        if (ctl[2 * b + 1] == 0u) continue;
        const int rc = one_by_one(b, b + 1);
        if (rc && rc != ET_ERR_BAD_DATA) return rc;  // bad data: states_host[b].bad_input is set, reported below
    }
    for (int64_t b = 0; b < batch; ++b)
        if (states_host[b].bad_input) return ET_ERR_BAD_DATA;
    return ET_OK;
}

extern "C" int et_kmeans_fit(const float *X, int64_t N, int d, int K, int max_iter, float tol, float *centroids,
                             int64_t *labels, float *trace, et_kmeans_state *state_host,
                             et_kmeans_timing *timing_host, void *workspace, size_t workspace_bytes,
                             et_stream_t stream) {
    if (!km_dims_ok(d, K) || N < 1 || !X || !centroids || !state_host || max_iter < 1) return ET_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < et_kmeans_workspace_bytes(N, d, K)) return ET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const KmWorkspace w = km_carve(workspace, N, d, K);
    // Timing (optional): HIP events around a SAMPLE of the assign launches -- the first one (plain exact scan) and
    // every kTimeEvery-th of the others (filter kernel).  An event record between two kernels costs a ~5 us
    // dispatch gap on each side, so timing every launch would slow the loop it measures by ~15 %.
    constexpr int kTimeEvery = 8;
    auto timed = [&](int it) { return timing_host && (it == 0 || it % kTimeEvery == 1); };
    // timing events belong to the device that is current when they are created: one cached set per (host thread, device)
    int dev_id = 0;
    ET_HIP_TRY(hipGetDevice(&dev_id));
    struct EventHolder {  // destroyed with the host thread
        std::vector<std::vector<hipEvent_t>> v;
        ~EventHolder() {
            for (auto &dev_events : v)
                for (hipEvent_t e : dev_events) (void)hipEventDestroy(e);
        }
    };
    static thread_local EventHolder per_thread;
    std::vector<std::vector<hipEvent_t>> &per_device_events = per_thread.v;
    if ((int)per_device_events.size() <= dev_id) per_device_events.resize(dev_id + 1);
    std::vector<hipEvent_t> &events = per_device_events[dev_id];
    if (timing_host) {
        while ((int)events.size() < 2 * max_iter) {
            hipEvent_t e;
            ET_HIP_TRY(hipEventCreate(&e));
            events.push_back(e);
        }
    }
    // The reference synchronises every iteration (error <= tol on the host, kmeans.py:239).  Here convergence
    // lives on the device: once state->done is set the remaining launches are no-ops.  The host never waits
    // for it inside the loop (et_hostring.h): the queue stays at most kSlots * kEvery iterations ahead.
    constexpr int kEvery = 4;
    int rc = ET_OK;
    StateRing *ring = StateRing::get(&rc);
    if (!ring) return rc;
    rc = et_kmeans_scan(X, N, d, w.state, stream);
    if (!rc) rc = et_kmeans_begin(w.state, N, centroids, d, K, stream);
    if (rc) return rc;
    // Without a trace the inertia of an iteration is not an output (kmeans.py:234 only prints it and keeps the last
    // one): the assignment kernels skip the fp64 similarity sums and the inertia of the LAST assignment is evaluated
    // by one extra pass after the loop -- the same bits as the per-iteration sum would have given.
    const bool want_sim = trace != nullptr;
    int launched = 0;
    bool done = false;
    // shards the matrix-core filter takes: the chained form (kmeans_lloyd_chain_kernel) -- every launch applies the
    // previous iteration's update in its prologue, in every workgroup; one more update after the loop.  (A one-launch
    // form with a ticketed fold + update in the last workgroup served shards <= 131072 points until its serial tail
    // lost to this prologue: 19.8 against 16.8 us per iteration at N = 1e5.)
    const bool chained = km_use_filter(X, N, d, K, w.labels_u8);
    // ... as ONE persistent launch for all iterations (kmeans_lloyd_persist_kernel); one launch per iteration
    // (km_chain_run, the form the sharded loop uses) if its grid barrier timed out or ET_KMEANS_LOOP=chain asks for it
    bool persisted = false;
    if (chained && km_persist_wanted(N)) {
        bool aborted = false;
        rc = km_persist_run(X, N, d, K, max_iter, tol, centroids, w.labels_u8, trace, w.state, (long long *)w.partials, w, st,
                            timing_host ? events[0] : nullptr, timing_host ? events[1] : nullptr, &aborted, nullptr);
        if (rc) return rc;
        persisted = !aborted;
    }
    if (chained && !persisted) {
        rc = km_chain_run(X, N, d, K, max_iter, tol, centroids, w.labels_u8, trace, w.state, (long long *)w.partials, w, st,
                          ChainHook{}, timing_host ? &events : nullptr, kTimeEvery, &launched);
        if (rc) return rc;
    }
    if (!chained) {
        ET_HIP_TRY(hipMemsetAsync(w.ticket, 0, sizeof(unsigned), st));
        ET_HIP_TRY(hipMemsetAsync(w.acc_lanes, 0, sizeof(long long) * km_plen(d, K) * 16, st));
    }
    for (int it = 0; !chained && it < max_iter && !done; ++it) {
        rc = assign_accumulate_impl(X, N, d, K, w.state, centroids, nullptr, w.labels_u8, (int64_t *)w.partials, workspace,
                                    workspace_bytes, st, timed(it) ? events[2 * it] : nullptr,
                                    timed(it) ? events[2 * it + 1] : nullptr, true, tol, trace, want_sim);
        if (rc) return rc;
        launched = it + 1;
        if (launched % kEvery == 0) {
            rc = ring->post(w.state, st, &done);
            if (rc) return rc;
        }
        ring->poll(&done);
    }
    if (!want_sim && !chained) {
        ET_HIP_TRY(hipMemsetAsync(w.sim_total, 0, 2 * sizeof(long long), st));
        const size_t ilds = sizeof(float) * (size_t)K * cpitch_host(d);
        const int igrid = km_grid(N);
        if (d == 6)
            hipLaunchKernelGGL((kmeans_inertia_kernel<6>), dim3(igrid), dim3(kKmThreads), ilds, st, X, N, d, K,
                               (const float *)w.last, (const uint8_t *)w.labels_u8, w.sim_total);
        else
            hipLaunchKernelGGL((kmeans_inertia_kernel<0>), dim3(igrid), dim3(kKmThreads), ilds, st, X, N, d, K,
                               (const float *)w.last, (const uint8_t *)w.labels_u8, w.sim_total);
        hipLaunchKernelGGL(kmeans_inertia_finish_kernel, dim3(1), dim3(64), 0, st, w.state, (const float *)w.last, d, K,
                           (const long long *)w.sim_total);
        ET_LAUNCH_CHECK();
    }
    if (labels) {  // (NULL: the caller only wants the centroids)
        rc = et_kmeans_labels_i64(w.labels_u8, N, labels, stream);
        if (rc) return rc;
    }
    ET_HIP_TRY(hipMemcpyAsync(state_host, w.state, sizeof(et_kmeans_state), hipMemcpyDeviceToHost, st));
    ET_HIP_TRY(hipStreamSynchronize(st));
    if (timing_host && persisted) {
        // one launch ran every assignment (the exact first pass included) and every update of the fit
        float ms = 0.f;
        ET_HIP_TRY(hipEventElapsedTime(&ms, events[0], events[1]));
        timing_host->assign_ms = (double)ms;
        timing_host->assign_launches = 1;
        timing_host->first_assign_ms = 0.0;
        timing_host->iterations = state_host->iter;
    } else if (timing_host) {
        // launches after convergence are no-ops (a few microseconds); count only the working ones.  The first
        // launch of a fit is the plain exact scan with full accumulation, the others the filter kernel.
        const int worked = (int)(state_host->iter < launched ? state_host->iter : launched);
        double total = 0.0, first = 0.0;
        int samples = 0;
        for (int it = 0; it < worked; ++it) {
            if (!timed(it)) continue;
            if (it > 0 && it + kTimedRun - 1 >= worked) continue;  // (a run that the fit's end cut short has no end event)
            float ms = 0.f;
            ET_HIP_TRY(hipEventElapsedTime(&ms, events[2 * it], events[2 * it + 1]));
            if (it == 0) {
                first = (double)ms;
            } else {
                total += (double)ms;
                samples += kTimedRun;
            }
        }
        timing_host->assign_ms = total;
        timing_host->assign_launches = samples;
        timing_host->first_assign_ms = first;
        timing_host->iterations = samples;
    }
    return state_host->bad_input ? ET_ERR_BAD_DATA : ET_OK;
}
