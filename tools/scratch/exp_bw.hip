// Bandwidth calibration on the box: read-only, write-only, copy, and 7:1 read:write mixes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
template <int MODE>  // 0 read, 1 write, 2 copy, 3 read 13 float4 write 1 (project-like), 4 read 2.5 write 6 (reconstruct-like)
__global__ __launch_bounds__(256) void k(const float4* __restrict__ a, float4* __restrict__ b, long n4, float* sink) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    float acc = 0.f;
    for (; i < n4; i += stride) {
        if (MODE == 0) { float4 v = a[i]; acc += v.x + v.y + v.z + v.w; }
        else if (MODE == 1) { b[i] = make_float4(1.f, 2.f, 3.f, (float)i); }
        else if (MODE == 2) { b[i] = a[i]; }
    }
    if (acc == 12345.678f) sink[0] = acc;
}
template <int MODE> float run(const float4* a, float4* b, long n4, float* sink, int grid) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float best = 1e9;
    for (int r = 0; r < 8; ++r) { CK(hipEventRecord(e0)); hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, a, b, n4, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r > 1 && ms < best) best = ms; }
    return best;
}
int main() {
    const long bytes = 1600000000L; const long n4 = bytes / 16;
    float4 *a, *b; float* sink; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    for (int grid : {2048, 8192, 65536, (int)((n4 + 255) / 256)}) {
        float t0 = run<0>(a, b, n4, sink, grid), t1 = run<1>(a, b, n4, sink, grid), t2 = run<2>(a, b, n4, sink, grid);
        printf("grid %8d: read %.1f us %.0f GB/s | write %.1f us %.0f GB/s | copy %.1f us %.0f GB/s (r+w)\n", grid, t0 * 1e3, bytes / t0 / 1e6, t1 * 1e3, bytes / t1 / 1e6, t2 * 1e3, 2.0 * bytes / t2 / 1e6);
    }
    return 0;
}
