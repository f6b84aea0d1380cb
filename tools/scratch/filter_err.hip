// Development aid: checks the D layout of v_mfma_f32_32x32x16_f16 and the error of the split-f16 similarity
// against fp64.  hipcc --offload-arch=gfx950 -O2 filter_err.hip -o filter_err && ./filter_err
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void split_f16(float a, float b, unsigned &hi, unsigned &lo) {
    const auto h = __builtin_amdgcn_cvt_pkrtz(a, b);
    const float ra = a - (float)h[0], rb = b - (float)h[1];
    const auto l = __builtin_amdgcn_cvt_pkrtz(ra, rb);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

// c: 32 clusters x 8 floats (c0..c5, bn, -); x: 32 points x 6; out: [row][col] = t'(cluster row, point col), raw rows
__global__ void k(const float *c, const float *x, float sg, float *out, float *raw) {
    const int lane = threadIdx.x, half = lane >> 5, col = lane & 31;
    unsigned ch[3], cl[3];
    for (int p = 0; p < 3; ++p) split_f16(2.f * sg * c[col * 8 + 2 * p], 2.f * sg * c[col * 8 + 2 * p + 1], ch[p], cl[p]);
    const float nb = -c[col * 8 + 6] * sg * sg;
    const auto nh = __builtin_amdgcn_cvt_pkrtz(nb, 0.f);
    const unsigned bnd = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(nb, (nb - (float)nh[0]) * 1024.0f));
    u32x4 a1, a2;
    if (half == 0) { a1 = u32x4{ch[0], ch[1], ch[2], ch[0]}; a2 = u32x4{ch[1], ch[2], bnd, 0u}; }
    else           { a1 = u32x4{cl[0], cl[1], cl[2], cl[0]}; a2 = u32x4{cl[1], cl[2], 0u, 0u}; }
    unsigned w[6];
    for (int p = 0; p < 3; ++p) split_f16(x[col * 6 + 2 * p] * sg, x[col * 6 + 2 * p + 1] * sg, w[p], w[3 + p]);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a1), __builtin_bit_cast(f16x8, u32x4{w[0], w[1], w[2], w[3]}), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a2), __builtin_bit_cast(f16x8, u32x4{w[4], w[5], 0x14003c00u, 0u}), acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        const int row = 8 * (r >> 2) + 4 * half + (r & 3);  // assumed layout
        out[row * 32 + col] = acc[r];
        raw[(half * 16 + r) * 32 + col] = acc[r];
    }
}

int main() {
    srand(1);
    double worst = 0, worst_abs = 0;
    int layout_bad = 0;
    for (int trial = 0; trial < 200; ++trial) {
        const float scale = trial < 100 ? 1.f : powf(2.f, (float)(rand() % 24 - 12));  // typical magnitude relative to the max
        std::vector<float> c(32 * 8), x(32 * 6);
        double mx = 0;
        for (int j = 0; j < 32; ++j) {
            float bn = 0.f;
            for (int i = 0; i < 6; ++i) {
                float v = (float)((rand() / (double)RAND_MAX * 2 - 1) * 3.0) * (j == 0 ? 1.f : scale);
                c[j * 8 + i] = v; bn = bn + v * v; mx = fmax(mx, fabs(v));
            }
            c[j * 8 + 6] = bn;
        }
        for (int n = 0; n < 32; ++n)
            for (int i = 0; i < 6; ++i) {
                float v = (float)((rand() / (double)RAND_MAX * 2 - 1) * 4.0) * (n == 0 ? 1.f : scale);
                x[n * 6 + i] = v; mx = fmax(mx, fabs(v));
            }
        int e; frexp(mx, &e);  // mx < 2^e
        const float sg = ldexpf(1.f, 5 - e);
        float *dc, *dx, *dout, *draw;
        hipMalloc(&dc, c.size() * 4); hipMalloc(&dx, x.size() * 4); hipMalloc(&dout, 1024 * 4); hipMalloc(&draw, 1024 * 4);
        hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dc, dx, sg, dout, draw);
        std::vector<float> out(1024);
        hipMemcpy(out.data(), dout, 4096, hipMemcpyDeviceToHost);
        hipFree(dc); hipFree(dx); hipFree(dout); hipFree(draw);
        for (int j = 0; j < 32; ++j)
            for (int n = 0; n < 32; ++n) {
                double g = 0, cn = 0, xn = 0;
                for (int i = 0; i < 6; ++i) {
                    g += 2.0 * c[j * 8 + i] * (double)x[n * 6 + i];
                    cn += (double)c[j * 8 + i] * c[j * 8 + i]; xn += (double)x[n * 6 + i] * x[n * 6 + i];
                }
                g = (g - (double)c[j * 8 + 6]) * sg * sg;
                const double rr = (sqrt(cn) + sqrt(xn)) * sg;
                const double err = fabs(out[j * 32 + n] - g);
                const double bound = ldexp(rr * rr, -17) + ldexp(rr, -21) + ldexp(1.0, -34);  // ~E2 (2^-17.5, 2^-21.7)
                if (err > 1e-2 * (rr * rr + 1e-30) && err > 1e-6) ++layout_bad;
                worst = fmax(worst, err / bound);
                worst_abs = fmax(worst_abs, err);
            }
    }
    printf("layout mismatches %d   worst err/E2 %.4f   worst abs err %.3e\n", layout_bad, worst, worst_abs);
    return 0;
}
