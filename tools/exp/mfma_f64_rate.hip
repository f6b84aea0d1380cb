// Issue rate of the two fp64 matrix instructions of gfx950, and of v_cvt_f64_f32 / v_fma_f64, one wavefront per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_f64_rate.hip -o /tmp/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k16(double *out, int iters) {
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.y + a2.z + a3.w;
}
__global__ void k4(double *out, int iters) {
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a3, 0, 0, 0);
        a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a4, 0, 0, 0);
        a5 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a5, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a6, 0, 0, 0);
        a7 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a7, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void kcvt(double *out, int iters, const float *in) {
    float f[8];
    for (int q = 0; q < 8; ++q) f[q] = in[threadIdx.x + 64 * q];
    double s = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            double dq;
            asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(dq) : "v"(f[q]));
            asm volatile("" ::"v"(dq));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void kfma(double *out, int iters) {
    double a[8];
    for (int q = 0; q < 8; ++q) a[q] = threadIdx.x + q;
    double x = 1.0000001, y = 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = __builtin_fma(a[q], x, y);
    }
    double s = 0;
    for (int q = 0; q < 8; ++q) s += a[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// a wavefront of matrix instructions beside a wavefront of fp32 vector work on the same SIMD: do they overlap?
__global__ void kmix(double *out, int iters) {
    const int w = threadIdx.x >> 6;
    if (w < 4) {
        d4 a0 = {0, 0, 0, 0}, a1 = a0;
        double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.y;
    } else {
        float a[8];
        for (int q = 0; q < 8; ++q) a[q] = threadIdx.x + q;
        float x = 1.0000001f, y = 1e-9f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 8; ++q) a[q] = __builtin_fmaf(a[q], x, y);
        }
        float s = 0;
        for (int q = 0; q < 8; ++q) s += a[q];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}
template <class F>
float timeit(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}
int main() {
    double *out;
    float *in;
    hipMalloc(&out, 1 << 24);
    hipMalloc(&in, 1 << 16);
    hipMemset(in, 0, 1 << 16);
    const int iters = 20000, grid = 256;
    // 256 threads = one wavefront per SIMD of a CU
    float t16 = timeit([&] { k16<<<grid, 256>>>(out, iters); });
    float t4 = timeit([&] { k4<<<grid, 256>>>(out, iters); });
    float tc = timeit([&] { kcvt<<<grid, 256>>>(out, iters, in); });
    float tf = timeit([&] { kfma<<<grid, 256>>>(out, iters); });
    float tm = timeit([&] { kmix<<<grid, 512>>>(out, iters); });
    float tm4 = timeit([&] { kmix<<<grid, 256>>>(out, iters); });
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("clock %d kHz\n", clk);
    printf("v_mfma_f64_16x16x4: %.3f ms for %d x 4 per wavefront: %.1f ns each (1024 FMA)\n", t16, iters, t16 * 1e6 / (iters * 4));
    printf("v_mfma_f64_4x4x4_4b: %.3f ms for %d x 8 per wavefront: %.1f ns each (256 FMA)\n", t4, iters, t4 * 1e6 / (iters * 8));
    printf("v_cvt_f64_f32: %.1f ns each;  v_fma_f64: %.1f ns each\n", tc * 1e6 / (iters * 8), tf * 1e6 / (iters * 8));
    printf("mix: 2 x 16x16x4 beside 32 v_fma_f32 of a second wavefront per SIMD: %.3f ms (matrix alone %.3f, per iteration %.1f ns vs %.1f)\n", tm, tm4,
           tm * 1e6 / iters, tm4 * 1e6 / iters);
    return 0;
}
