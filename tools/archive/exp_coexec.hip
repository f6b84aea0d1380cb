// Which VALU instruction classes issue while an fp64 MFMA of ANOTHER wavefront on the same SIMD is executing?
//   hipcc --offload-arch=gfx950 -O3 tools/exp_coexec.hip -o /tmp/exp_coexec && /tmp/exp_coexec
// One 768-thread workgroup per CU (three wavefronts per SIMD).  Wavefront slots 0 and 1 of a SIMD run a chain-free stream
// of MFMAs, slot 2 a stream of VALU instructions of one class; timed alone and in combination (one MFMA issuer per SIMD
// reaches half the matrix-pipe rate, two saturate it -- does the VALU wavefront still make progress then?).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MF, int VK>  // MF 0: f64 16x16x4, 1: f16 32x32x16;  VK 0: f32 fma, 1: cvt f32->f64, 2: dpp mov, 3: f64 fma, 4: cndmask
__global__ __launch_bounds__(768) void k(int mode, int iters, double *out) {
    const int slot = (threadIdx.x >> 6) >> 2;  // wavefronts 0-3 -> SIMD 0-3 first slot, 4-7 second, 8-11 third
    const bool mf = slot < 2;
    if (mode == 0 && slot != 0) return;
    if (mode == 1 && slot != 2) return;
    if (mode == 2 && slot == 1) return;
    if (mode == 3 && slot == 2) return;
    if (mf) {
        if (MF == 0) {
            f64x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;
            const double x = (double)threadIdx.x * 1e-3, y = x + 1.0;
            for (int i = 0; i < iters; ++i) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
            }
            out[blockIdx.x * 768 + threadIdx.x] = a0[0] + a1[1] + a2[2];
        } else {
            f32x16 a0, a1, a2;
            for (int r = 0; r < 16; ++r) a0[r] = a1[r] = a2[r] = 0.f;
            f16x8 x, y;
            for (int r = 0; r < 8; ++r) { x[r] = (_Float16)(threadIdx.x * 1e-3f); y[r] = (_Float16)1.0f; }
            for (int i = 0; i < iters; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, a2, 0, 0, 0);
            }
            out[blockIdx.x * 768 + threadIdx.x] = a0[0] + a1[1] + a2[2];
        }
    } else {
        float v[8];
        double dsum[8];
        for (int j = 0; j < 8; ++j) { v[j] = threadIdx.x + j; dsum[j] = j; }
        const float m = 1.0001f, c = 0.5f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (VK == 0) v[j] = fmaf(v[j], m, c);
                    if (VK == 1) { double d = (double)v[j]; asm volatile("" : "+v"(d)); dsum[j] = d; v[j] += 1.0f; }
                    if (VK == 2) v[j] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[j]), 0xB1, 0xF, 0xF, true)) ;
                    if (VK == 3) dsum[j] = fma(dsum[j], 1.0001, 0.5);
                    if (VK == 4) v[j] = v[(j + 1) & 7] > c ? v[j] : m;
                }
            }
        }
        double s = 0;
        for (int j = 0; j < 8; ++j) s += v[j] + dsum[j];
        out[blockIdx.x * 768 + threadIdx.x] = s;
    }
}

template <int MF, int VK>
static void run(const char *name) {
    double *out;
    (void)hipMalloc(&out, 256 * 768 * sizeof(double));
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int iters = 20000;
    float t[5];
    for (int mode = 0; mode < 5; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL((k<MF, VK>), dim3(256), dim3(768), 0, 0, mode, iters, out);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            float ms;
            (void)hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
        }
        t[mode] = best;
    }
    printf("%-26s 1 MFMA wf/SIMD %.3f | VALU wf %.3f | 1 MFMA + VALU %.3f | 2 MFMA wf/SIMD %.3f | 2 MFMA + VALU %.3f ms\n", name,
           t[0], t[1], t[2], t[3], t[4]);
    (void)hipFree(out);
}

int main() {
    run<0, 0>("f64 MFMA + f32 fma");
    run<0, 1>("f64 MFMA + cvt f32->f64");
    run<0, 2>("f64 MFMA + dpp mov");
    run<0, 3>("f64 MFMA + f64 fma");
    run<0, 4>("f64 MFMA + cmp/cndmask");
    run<1, 0>("f16 MFMA + f32 fma");
    run<1, 1>("f16 MFMA + cvt f32->f64");
    return 0;
}
