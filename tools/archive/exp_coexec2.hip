// Same-wavefront interleave: does a wavefront's own VALU work issue while its f64 MFMA executes?
// one wavefront per SIMD (256 threads per CU); per iteration 3 MFMAs and NV independent fp32 FMAs, in program order
// MFMA, NV/3 FMAs, MFMA, ... ; compare NV = 0 / 24 / 48 and the FMA-only loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NV, bool MF, int WAVES, int KIND>
__global__ __launch_bounds__(64 * WAVES) void k(int iters, double *out) {
    f64x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;
    const double x = (double)threadIdx.x * 1e-3, y = x + 1.0;
    f32x16 h0, h1, h2;
    for (int r = 0; r < 16; ++r) h0[r] = h1[r] = h2[r] = 0.f;
    f16x8 hx, hy;
    for (int r = 0; r < 8; ++r) { hx[r] = (_Float16)(threadIdx.x * 1e-3f); hy[r] = (_Float16)1.0f; }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x + j;
    const float m = 1.0001f, c = 0.5f;
    for (int i = 0; i < iters; ++i) {
        if (MF && KIND == 0) a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        if (MF && KIND == 1) h0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx, hy, h0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV / 3; ++j) v[j % 8] = fmaf(v[j % 8], m, c);
        if (MF && KIND == 0) a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
        if (MF && KIND == 1) h1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hy, hx, h1, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV / 3; ++j) v[(j + 3) % 8] = fmaf(v[(j + 3) % 8], m, c);
        if (MF && KIND == 0) a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
        if (MF && KIND == 1) h2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hx, hx, h2, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV / 3; ++j) v[(j + 5) % 8] = fmaf(v[(j + 5) % 8], m, c);
        __builtin_amdgcn_sched_barrier(0);
    }
    double s = a0[0] + a1[1] + a2[2] + h0[0] + h1[1] + h2[2];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}

template <int NV, bool MF, int WAVES, int KIND>
static float run() {
    double *out;
    (void)hipMalloc(&out, 256 * 64 * WAVES * sizeof(double));
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k<NV, MF, WAVES, KIND>), dim3(256), dim3(64 * WAVES), 0, 0, 20000, out);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    (void)hipFree(out);
    return best;
}

template <int KIND>
static void table(const char *name) {
    printf("%s, 4 wavefronts/SIMD, per iteration 3 MFMA (+ NV independent fp32 FMAs of the same wavefront):\n", name);
    const float m0 = run<0, true, 16, KIND>(), m24 = run<24, true, 16, KIND>(), m48 = run<48, true, 16, KIND>();
    const float v24 = run<24, false, 16, KIND>(), v48 = run<48, false, 16, KIND>();
    printf("  MFMA alone %.3f ms | FMA alone: 24 -> %.3f, 48 -> %.3f | together: 24 -> %.3f (sum %.3f), 48 -> %.3f (sum %.3f)\n", m0, v24, v48, m24,
           m0 + v24, m48, m0 + v48);
}

int main() {
    table<0>("v_mfma_f64_16x16x4_f64");
    table<1>("v_mfma_f32_32x32x16_f16");
    return 0;
}
