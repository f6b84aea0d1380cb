// Read-pattern experiment for the Lloyd kernel's memory side (run on the GPU box):
//     hipcc --offload-arch=gfx950 -O3 tools/exp_readpattern.hip -o /tmp/exp_readpattern && /tmp/exp_readpattern
// The filter kernel reads, per pass of a wavefront, 256 consecutive points of six coordinate rows (16 B per lane and
// row) + 4 label bytes per lane.  A loads-only copy of it runs as long as the full kernel (60 us at N = 1e7), i.e. the
// kernel is bound by how this pattern is served; the variants below look for a geometry the memory system likes better.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// generic kernel: WAVES_PER_WG wavefronts per workgroup; every wavefront handles passes g = wave_global, += total waves;
// DEPTH passes' loads are issued before the first is consumed
template <int THREADS, int DEPTH, int VEC>
__global__ __launch_bounds__(THREADS) void read_kernel(const float *__restrict__ X, const unsigned char *__restrict__ labels,
                                                       long long N, int persistent, unsigned *__restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long pts_per_pass = 64 * VEC;
    const long long n_groups = (N + pts_per_pass - 1) / pts_per_pass;
    const long long waves_total = (long long)gridDim.x * (THREADS / 64);
    float acc = 0.f;
    long long g = (long long)blockIdx.x * (THREADS / 64) + wave;
    const long long step = persistent ? waves_total : n_groups;  // non-persistent: exactly one pass per wavefront
    for (; g < n_groups; g += step * DEPTH) {
        float buf[DEPTH][6][VEC];
        unsigned lb[DEPTH];
#pragma unroll
        for (int dph = 0; dph < DEPTH; ++dph) {
            const long long gg = g + dph * step;
            const long long n = gg * pts_per_pass + (long long)lane * VEC;
            const bool ok = gg < n_groups && n < N;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                if (VEC == 4) {
                    const float4 v = ok ? *reinterpret_cast<const float4 *>(X + (long long)i * N + n) : make_float4(0, 0, 0, 0);
                    buf[dph][i][0] = v.x; buf[dph][i][1 % VEC] = v.y; buf[dph][i][2 % VEC] = v.z; buf[dph][i][3 % VEC] = v.w;
                } else if (VEC == 2) {
                    const float2 v = ok ? *reinterpret_cast<const float2 *>(X + (long long)i * N + n) : make_float2(0, 0);
                    buf[dph][i][0] = v.x; buf[dph][i][1 % VEC] = v.y;
                } else {
                    buf[dph][i][0] = ok ? X[(long long)i * N + n] : 0.f;
                }
            }
            lb[dph] = ok ? (VEC == 4 ? *reinterpret_cast<const unsigned *>(labels + n) : (unsigned)labels[n]) : 0u;
        }
#pragma unroll
        for (int dph = 0; dph < DEPTH; ++dph) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc += buf[dph][i][v];
            acc += (float)lb[dph];
        }
    }
    if (acc == 1.2345e-30f) sink[0] = 1;
}

template <int THREADS, int DEPTH, int VEC>
static void run(const char *name, const float *X, const unsigned char *lab, long long N, int grid, int persistent, unsigned *sink) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const long long n_groups = (N + 64 * VEC - 1) / (64 * VEC);
    if (!persistent) grid = (int)((n_groups + (THREADS / 64) * DEPTH - 1) / ((THREADS / 64) * DEPTH));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((read_kernel<THREADS, DEPTH, VEC>), dim3(grid), dim3(THREADS), 0, 0, X, lab, N, persistent, sink);
    CHECK(hipEventRecord(a));
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((read_kernel<THREADS, DEPTH, VEC>), dim3(grid), dim3(THREADS), 0, 0, X, lab, N, persistent, sink);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / reps;
    printf("%-44s grid %7d  %8.2f us  %7.0f GB/s\n", name, grid, us, 25.0 * N / us / 1e3);
}

int main(int argc, char **argv) {
    const long long N = argc > 1 ? atoll(argv[1]) : 10000000;
    float *X; unsigned char *lab; unsigned *sink;
    CHECK(hipMalloc(&X, sizeof(float) * 6 * N)); CHECK(hipMalloc(&lab, N + 64)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(X, 0, sizeof(float) * 6 * N)); CHECK(hipMemset(lab, 1, N + 64));
    int cus = 256;
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0)); cus = p.multiProcessorCount;
    printf("N = %lld, %d CUs\n", N, cus);
    run<1024, 1, 4>("A  1024 thr x 1/CU, persistent, depth 1", X, lab, N, cus, 1, sink);
    run<1024, 2, 4>("B  1024 thr x 1/CU, persistent, depth 2", X, lab, N, cus, 1, sink);
    run<1024, 1, 4>("C  1024 thr, one pass per wavefront", X, lab, N, 0, 0, sink);
    run<512, 1, 4>("D  512 thr x 2/CU, persistent", X, lab, N, 2 * cus, 1, sink);
    run<256, 1, 4>("E  256 thr x 4/CU, persistent", X, lab, N, 4 * cus, 1, sink);
    run<256, 1, 4>("F  256 thr x 8/CU, persistent", X, lab, N, 8 * cus, 1, sink);
    run<256, 1, 4>("G  256 thr, one pass per wavefront", X, lab, N, 0, 0, sink);
    run<256, 2, 4>("H  256 thr x 8/CU, persistent, depth 2", X, lab, N, 8 * cus, 1, sink);
    run<1024, 1, 2>("I  1024 thr x 1/CU, persistent, 8 B/lane", X, lab, N, cus, 1, sink);
    run<256, 1, 2>("J  256 thr x 8/CU, persistent, 8 B/lane", X, lab, N, 8 * cus, 1, sink);
    run<1024, 1, 4>("K  1024 thr x 2/CU, persistent", X, lab, N, 2 * cus, 1, sink);
    run<1024, 4, 4>("L  1024 thr x 1/CU, persistent, depth 4", X, lab, N, cus, 1, sink);
    return 0;
}
