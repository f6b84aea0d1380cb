// Stand-alone experiment: cost of the k-means arg-max loop alone (no accumulation), several structures.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

__device__ __forceinline__ void argmax4(const f2 (&xa)[6], const f2 (&xb)[6], const float* sC, int K, unsigned& packed, float& bsum) {
    f2 ana = {0.f, 0.f}, anb = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 6; ++i) { ana = ana + xa[i] * xa[i]; anb = anb + xb[i] * xb[i]; }
    const float4* s4 = reinterpret_cast<const float4*>(sC);
    float4 c0 = s4[0], c1 = s4[1];
    float b0, b1, b2, b3; int l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    for (int j = 0; j < K; ++j) {
        const float4 p0 = c0, p1 = c1;
        if (j + 1 < K) { c0 = s4[2 * j + 2]; c1 = s4[2 * j + 3]; }
        f2 ya = {0.f, 0.f}, yb = {0.f, 0.f};
        const float cc[6] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y};
#pragma unroll
        for (int i = 0; i < 6; ++i) { const f2 c = {cc[i], cc[i]}; ya = __builtin_elementwise_fma(xa[i], c, ya); yb = __builtin_elementwise_fma(xb[i], c, yb); }
        ya = ya * 2.0f; yb = yb * 2.0f; ya = ya - ana; yb = yb - anb;
        const f2 bn = {p1.z, p1.z}; ya = ya - bn; yb = yb - bn;
        if (j == 0) { b0 = ya.x; b1 = ya.y; b2 = yb.x; b3 = yb.y; }
        else {
            const bool t0 = ya.x > b0, t1 = ya.y > b1, t2 = yb.x > b2, t3 = yb.y > b3;
            b0 = t0 ? ya.x : b0; l0 = t0 ? j : l0; b1 = t1 ? ya.y : b1; l1 = t1 ? j : l1;
            b2 = t2 ? yb.x : b2; l2 = t2 ? j : l2; b3 = t3 ? yb.y : b3; l3 = t3 ? j : l3;
        }
    }
    packed = l0 | (l1 << 8) | (l2 << 16) | (l3 << 24); bsum = b0 + b1 + b2 + b3;
}

__device__ __forceinline__ void argmax4s(const float (&x)[4][6], const float* sC, int K, unsigned& packed, float& bsum) {
    float an[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) { an[v] = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) an[v] = an[v] + x[v][i] * x[v][i]; }
    const float4* s4 = reinterpret_cast<const float4*>(sC);
    float4 c0 = s4[0], c1 = s4[1];
    float b[4]; int l[4] = {0, 0, 0, 0};
    for (int j = 0; j < K; ++j) {
        const float4 p0 = c0, p1 = c1;
        if (j + 1 < K) { c0 = s4[2 * j + 2]; c1 = s4[2 * j + 3]; }
        const float cc[6] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y};
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float y = 0.f;
#pragma unroll
            for (int i = 0; i < 6; ++i) y = fmaf(x[v][i], cc[i], y);
            y = y * 2.0f; y = y - an[v]; y = y - p1.z;
            if (j == 0) b[v] = y; else { const bool t = y > b[v]; b[v] = t ? y : b[v]; l[v] = t ? j : l[v]; }
        }
    }
    packed = l[0] | (l[1] << 8) | (l[2] << 16) | (l[3] << 24); bsum = b[0] + b[1] + b[2] + b[3];
}

// fully unrolled K (compile time), no manual prefetch: the scheduler places the LDS reads
template <int KK>
__device__ __forceinline__ void argmax4u(const f2 (&xa)[6], const f2 (&xb)[6], const float* sC, unsigned& packed, float& bsum) {
    f2 ana = {0.f, 0.f}, anb = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 6; ++i) { ana = ana + xa[i] * xa[i]; anb = anb + xb[i] * xb[i]; }
    const float4* s4 = reinterpret_cast<const float4*>(sC);
    float b0, b1, b2, b3; int l0 = 0, l1 = 0, l2 = 0, l3 = 0;
#pragma unroll
    for (int j = 0; j < KK; ++j) {
        const float4 p0 = s4[2 * j], p1 = s4[2 * j + 1];
        f2 ya = {0.f, 0.f}, yb = {0.f, 0.f};
        const float cc[6] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y};
#pragma unroll
        for (int i = 0; i < 6; ++i) { const f2 c = {cc[i], cc[i]}; ya = __builtin_elementwise_fma(xa[i], c, ya); yb = __builtin_elementwise_fma(xb[i], c, yb); }
        ya = ya * 2.0f; yb = yb * 2.0f; ya = ya - ana; yb = yb - anb;
        const f2 bn = {p1.z, p1.z}; ya = ya - bn; yb = yb - bn;
        if (j == 0) { b0 = ya.x; b1 = ya.y; b2 = yb.x; b3 = yb.y; }
        else {
            const bool t0 = ya.x > b0, t1 = ya.y > b1, t2 = yb.x > b2, t3 = yb.y > b3;
            b0 = t0 ? ya.x : b0; l0 = t0 ? j : l0; b1 = t1 ? ya.y : b1; l1 = t1 ? j : l1;
            b2 = t2 ? yb.x : b2; l2 = t2 ? j : l2; b3 = t3 ? yb.y : b3; l3 = t3 ? j : l3;
        }
    }
    packed = l0 | (l1 << 8) | (l2 << 16) | (l3 << 24); bsum = b0 + b1 + b2 + b3;
}

// 8 points per lane (4 packed pairs): more independent chains per wave
__device__ __forceinline__ void argmax8(const f2 (&x)[4][6], const float* sC, int K, unsigned (&packed)[2], float& bsum) {
    f2 an[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) { an[p] = f2{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 6; ++i) an[p] = an[p] + x[p][i] * x[p][i]; }
    const float4* s4 = reinterpret_cast<const float4*>(sC);
    float4 c0 = s4[0], c1 = s4[1];
    f2 b[4]; int l[8] = {0,0,0,0,0,0,0,0};
    for (int j = 0; j < K; ++j) {
        const float4 p0 = c0, p1 = c1;
        if (j + 1 < K) { c0 = s4[2 * j + 2]; c1 = s4[2 * j + 3]; }
        const float cc[6] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y};
        const f2 bn = {p1.z, p1.z};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            f2 y = {0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 6; ++i) { const f2 c = {cc[i], cc[i]}; y = __builtin_elementwise_fma(x[p][i], c, y); }
            y = y * 2.0f; y = y - an[p]; y = y - bn;
            if (j == 0) b[p] = y;
            else { const bool t0 = y.x > b[p].x, t1 = y.y > b[p].y; b[p].x = t0 ? y.x : b[p].x; l[2*p] = t0 ? j : l[2*p]; b[p].y = t1 ? y.y : b[p].y; l[2*p+1] = t1 ? j : l[2*p+1]; }
        }
    }
    packed[0] = l[0] | (l[1] << 8) | (l[2] << 16) | (l[3] << 24); packed[1] = l[4] | (l[5] << 8) | (l[6] << 16) | (l[7] << 24);
    bsum = b[0].x + b[0].y + b[1].x + b[1].y + b[2].x + b[2].y + b[3].x + b[3].y;
}

// V=0: one group per thread; V=1: grid-stride; V=2: grid-stride with next-group prefetch; V=3: like 0 but no compute (copy floor)
template <int V>
__global__ __launch_bounds__(256) void k(const float* __restrict__ X, long N, int K, const float* __restrict__ cen8, unsigned* __restrict__ out, float* __restrict__ acc) {
    extern __shared__ __attribute__((aligned(16))) float sC[];
    for (int i = threadIdx.x; i < K * 8; i += 256) sC[i] = cen8[i];
    __syncthreads();
    const long ngroups = N / 4;
    const long stride = (long)gridDim.x * 256;
    float total = 0.f;
    long g = (long)blockIdx.x * 256 + threadIdx.x;
    if (V == 0 || V == 3 || V == 4 || V == 5) {
        if (g >= ngroups) return;
        f2 xa[6], xb[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) { const float4 v = *reinterpret_cast<const float4*>(X + (long)i * N + g * 4); xa[i] = f2{v.x, v.y}; xb[i] = f2{v.z, v.w}; }
        unsigned p; float b;
        if (V == 0) argmax4(xa, xb, sC, K, p, b);
        else if (V == 5) argmax4u<20>(xa, xb, sC, p, b);
        else if (V == 4) { float xs[4][6];
#pragma unroll
            for (int i = 0; i < 6; ++i) { xs[0][i] = xa[i].x; xs[1][i] = xa[i].y; xs[2][i] = xb[i].x; xs[3][i] = xb[i].y; }
            argmax4s(xs, sC, K, p, b); }
        else { p = 0; b = 0; for (int i = 0; i < 6; ++i) b += xa[i].x + xa[i].y + xb[i].x + xb[i].y; }
        out[g] = p; total = b;
    } else if (V == 1) {
        for (; g < ngroups; g += stride) {
            f2 xa[6], xb[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) { const float4 v = *reinterpret_cast<const float4*>(X + (long)i * N + g * 4); xa[i] = f2{v.x, v.y}; xb[i] = f2{v.z, v.w}; }
            unsigned p; float b; argmax4(xa, xb, sC, K, p, b); out[g] = p; total += b;
        }
    } else {
        float4 nx[6];
        if (g < ngroups) {
#pragma unroll
            for (int i = 0; i < 6; ++i) nx[i] = *reinterpret_cast<const float4*>(X + (long)i * N + g * 4);
        }
        for (; g < ngroups; g += stride) {
            f2 xa[6], xb[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) { xa[i] = f2{nx[i].x, nx[i].y}; xb[i] = f2{nx[i].z, nx[i].w}; }
            const long gn = g + stride;
            if (gn < ngroups) {
#pragma unroll
                for (int i = 0; i < 6; ++i) nx[i] = *reinterpret_cast<const float4*>(X + (long)i * N + gn * 4);
            }
            unsigned p; float b; argmax4(xa, xb, sC, K, p, b); out[g] = p; total += b;
        }
    }
    if (total == 123.456f) acc[0] = total;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ unsigned other_half(unsigned v, int half) { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return half ? r[0] : r[1]; }
// M=0: full arg-max; M=1: MFMA only (sum acc); M=2: no MFMA (loads + fake argmax)
template <int M>
__global__ __launch_bounds__(256) void km(const float* __restrict__ X, long N, int K, const float* __restrict__ cen8, unsigned* __restrict__ out, float* __restrict__ accout) {
    const int lane = threadIdx.x & 63, half = lane >> 5, col = lane & 31;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, bn = __int_as_float(0x7f800000);
    if (col < K) { bn = cen8[col * 8 + 6]; a0 = 2.f * cen8[col * 8 + half]; a1 = 2.f * cen8[col * 8 + 2 + half]; a2 = 2.f * cen8[col * 8 + 4 + half]; }
    const float a3 = half ? -bn : 1.0f;
    const int nblk = (K + 7) >> 3;
    const long n_groups = (N + 127) / 128;
    const int wave = threadIdx.x >> 6;
    float total = 0.f;
    for (long g = (long)blockIdx.x * 4 + wave; g < n_groups; g += (long)gridDim.x * 4) {
        const long n = g * 128 + 4 * col;
        float4 v[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = *reinterpret_cast<const float4*>(X + (long)i * N + n);
        unsigned packed = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float x[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) x[i] = q == 0 ? v[i].x : (q == 1 ? v[i].y : (q == 2 ? v[i].z : v[i].w));
            float an = 0.f;
#pragma unroll
            for (int i = 0; i < 6; ++i) an = an + x[i] * x[i];
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (M != 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, half ? x[1] : x[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, half ? x[3] : x[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, half ? x[5] : x[4], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, half ? 1.0f : -an, acc, 0, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = x[r % 6] * a0 + an * (float)r;
            }
            if (M == 1) { float s = 0; for (int r = 0; r < 16; ++r) s += acc[r]; total += s; continue; }
            float bv = __int_as_float(0xff800000); int lb = 0;
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) if (blk < nblk) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float y = acc[4 * blk + e]; const bool t = y > bv; bv = t ? y : bv; lb = t ? (e + 8 * blk + 4 * half) : lb; }
            }
            const float pv = __uint_as_float(other_half(__float_as_uint(bv), half));
            const int pl = (int)other_half((unsigned)lb, half);
            const bool tp = (pv > bv) || (pv == bv && pl < lb);
            bv = tp ? pv : bv; lb = tp ? pl : lb;
            packed |= (unsigned)lb << (8 * q); total += bv;
        }
        if (half == 0) out[g * 32 + col] = packed;
    }
    if (total == 123.456f) accout[0] = total;
}
template <int M> float runm(const float* X, long N, int K, const float* cen, unsigned* out, float* acc, int grid) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); float best = 1e9;
    for (int r = 0; r < 6; ++r) { CK(hipEventRecord(a)); hipLaunchKernelGGL((km<M>), dim3(grid), dim3(256), 0, 0, X, N, K, cen, out, acc); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r > 0 && ms < best) best = ms; }
    return best;
}

__global__ __launch_bounds__(256) void k8(const float* __restrict__ X, long N, int K, const float* __restrict__ cen8, unsigned* __restrict__ out, float* __restrict__ acc) {
    extern __shared__ __attribute__((aligned(16))) float sC[];
    for (int i = threadIdx.x; i < K * 8; i += 256) sC[i] = cen8[i];
    __syncthreads();
    const long g = (long)blockIdx.x * 256 + threadIdx.x;  // group of 8 points
    if (g * 8 >= N) return;
    f2 x[4][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(X + (long)i * N + g * 8);
        const float4 w = *reinterpret_cast<const float4*>(X + (long)i * N + g * 8 + 4);
        x[0][i] = f2{v.x, v.y}; x[1][i] = f2{v.z, v.w}; x[2][i] = f2{w.x, w.y}; x[3][i] = f2{w.z, w.w};
    }
    unsigned p[2]; float b; argmax8(x, sC, K, p, b);
    out[2 * g] = p[0]; out[2 * g + 1] = p[1];
    if (b == 123.456f) acc[0] = b;
}

template <int V> float run(const float* X, long N, int K, const float* cen, unsigned* out, float* acc, int grid) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int r = 0; r < 6; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<V>), dim3(grid), dim3(256), K * 8 * 4, 0, X, N, K, cen, out, acc);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r > 0 && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    long N = argc > 1 ? atol(argv[1]) : 10000000; int K = 20;
    float *X, *cen, *acc; unsigned* out;
    CK(hipMalloc(&X, N * 6 * 4)); CK(hipMalloc(&cen, K * 8 * 4)); CK(hipMalloc(&out, N)); CK(hipMalloc(&acc, 4));
    std::vector<float> h(N * 6); for (long i = 0; i < N * 6; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 100.f - 10.f;
    CK(hipMemcpy(X, h.data(), N * 6 * 4, hipMemcpyHostToDevice));
    std::vector<float> c(K * 8); for (int i = 0; i < K * 8; ++i) c[i] = (float)((i * 40503u) % 1999) / 100.f - 10.f;
    CK(hipMemcpy(cen, c.data(), K * 8 * 4, hipMemcpyHostToDevice));
    long ngroups = N / 4; int full = (int)((ngroups + 255) / 256);
    printf("N=%ld  bytes=%.0f MB\n", N, N * 24 / 1e6);
    float t;
    t = run<3>(X, N, K, cen, out, acc, full); printf("V3 load-only 1grp/thread grid=%d: %.1f us  %.0f GB/s\n", full, t * 1e3, N * 24 / t / 1e6);
    t = run<0>(X, N, K, cen, out, acc, full); printf("V0 argmax 1grp/thread grid=%d: %.1f us  %.0f GB/s\n", full, t * 1e3, N * 24 / t / 1e6);
    { hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); float best = 1e9; int grid8 = (int)((N / 8 + 255) / 256);
      for (int r = 0; r < 6; ++r) { CK(hipEventRecord(a)); hipLaunchKernelGGL(k8, dim3(grid8), dim3(256), K * 8 * 4, 0, X, N, K, cen, out, acc); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r > 0 && ms < best) best = ms; }
      printf("V8 argmax 8pts/lane packed grid=%d: %.1f us  %.0f GB/s\n", grid8, best * 1e3, N * 24 / best / 1e6); }
    t = run<5>(X, N, K, cen, out, acc, full); printf("V5 argmax K=20 unrolled grid=%d: %.1f us  %.0f GB/s\n", full, t * 1e3, N * 24 / t / 1e6);
    t = run<4>(X, N, K, cen, out, acc, full); printf("V4 argmax scalar 1grp/thread grid=%d: %.1f us  %.0f GB/s\n", full, t * 1e3, N * 24 / t / 1e6);
    for (int grid : {4096}) {
        float t0 = runm<0>(X, N, K, cen, out, acc, grid), t1 = runm<1>(X, N, K, cen, out, acc, grid), t2 = runm<2>(X, N, K, cen, out, acc, grid);
        printf("MFMA grid=%d: full %.1f us | mfma-only %.1f us | no-mfma %.1f us\n", grid, t0 * 1e3, t1 * 1e3, t2 * 1e3);
    }
    for (int grid : {4096}) {
        t = run<1>(X, N, K, cen, out, acc, grid); printf("V1 grid-stride grid=%d: %.1f us  %.0f GB/s\n", grid, t * 1e3, N * 24 / t / 1e6);
        t = run<2>(X, N, K, cen, out, acc, grid); printf("V2 grid-stride+prefetch grid=%d: %.1f us  %.0f GB/s\n", grid, t * 1e3, N * 24 / t / 1e6);
    }
    return 0;
}
