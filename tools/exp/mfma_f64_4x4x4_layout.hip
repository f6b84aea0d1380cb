// Operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950, found by experiment: A = 1 in one lane, B = a distinct
// value per lane; every non-zero D lane names the B lane it was paired with.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double *out) {
    const int lane = threadIdx.x;
    for (int e = 0; e < 64; ++e) {
        const double a = lane == e ? 1.0 : 0.0, b = (double)(lane + 1);
        const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        out[e * 64 + lane] = d;
    }
}
int main() {
    double *d, h[64 * 64];
    hipMalloc(&d, sizeof h);
    probe<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int e = 0; e < 64; ++e) {
        printf("A lane %2d:", e);
        for (int l = 0; l < 64; ++l)
            if (h[e * 64 + l] != 0.0) printf("  D[%2d]=B[%2d]", l, (int)h[e * 64 + l] - 1);
        printf("\n");
    }
    return 0;
}
