// Can consecutive launches of one stream overlap (no barrier bit: hipExtAnyOrderLaunch) so that launch t+1's workgroups
// are already resident and spinning on a completion counter when launch t's last workgroup finishes?  Emulates the
// chained Lloyd loop: 256 one-per-CU workgroups (768 threads, 51 KB LDS), each launch = wait for the previous launch's
// counter, acquire, fold a small table, `work` FMAs per thread, device atomics onto the next table, release, count.
//   build: hipcc --offload-arch=gfx950 -O3 tools/exp_overlap.hip -o gpurun_out/exp_overlap
//   every spin carries a timeout (s_memrealtime, 100 MHz): a wrong assumption ends the run, it cannot hang the GPU.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

constexpr int kLaunches = 100;
constexpr int kTable = 142 * 16;

struct Ctl {
    unsigned done[kLaunches + 4];
    unsigned timeout;
    unsigned pad[3];
    unsigned long long first_start[kLaunches], last_end[kLaunches], first_go[kLaunches];
    unsigned xcc[kLaunches][256];
};

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }

template <int MODE>  // 0: ordinary launches, 1: overlapped launches with a spin on the previous counter
__global__ __launch_bounds__(768) void chain(Ctl *c, int t, int work, unsigned long long *tables, float *sink) {
    extern __shared__ unsigned char lds[];
    __shared__ unsigned long long sSum;
    unsigned long long t0 = 0;
    if (threadIdx.x == 0) {
        t0 = now();
        sSum = 0;
        atomicMin(&c->first_start[t], t0);
        c->xcc[t][blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;  // XCC_ID
        if (MODE == 1 && t > 0) {
            while (__hip_atomic_load(&c->done[t - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
                if (now() - t0 > 5000000ull) {  // 50 ms
                    c->timeout = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        atomicMin(&c->first_go[t], now());
    }
    __syncthreads();
    if (MODE == 1) __atomic_thread_fence(__ATOMIC_ACQUIRE);  // agent scope: invalidate what this XCD may hold of the tables
    // fold: every workgroup reads the previous launch's table
    const unsigned long long *rd = tables + (size_t)(t % 3) * kTable;
    unsigned long long *wr = tables + (size_t)((t + 1) % 3) * kTable;
    unsigned long long *zr = tables + (size_t)((t + 2) % 3) * kTable;
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < kTable; i += blockDim.x) s += rd[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sSum, s);
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < kTable; i += blockDim.x) zr[i] = 0;
    __syncthreads();
    float v = (float)threadIdx.x + (float)(sSum & 0xff);
    for (int i = 0; i < work; ++i) v = fmaf(v, 1.0001f, 0.5f);
    if (v == 12345.678f) sink[0] = v;
    lds[threadIdx.x] = (unsigned char)v;
    // deltas: 142 device-scope atomics per workgroup
    if (threadIdx.x < 142) atomicAdd(&wr[threadIdx.x * 16 + (blockIdx.x & 15)], (unsigned long long)(t + 1));
    __syncthreads();
    if (threadIdx.x == 0) {
        if (MODE == 1) __atomic_thread_fence(__ATOMIC_RELEASE);
        __hip_atomic_fetch_add(&c->done[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicMax(&c->last_end[t], now());
    }
}

static void run(int mode, int work, int flags) {
    Ctl *c;
    unsigned long long *tables;
    float *sink;
    (void)hipMalloc(&c, sizeof(Ctl));
    (void)hipMalloc(&tables, sizeof(unsigned long long) * kTable * 3);
    (void)hipMalloc(&sink, 64);
    std::vector<unsigned char> init(sizeof(Ctl), 0);
    Ctl *h = reinterpret_cast<Ctl *>(init.data());
    for (int t = 0; t < kLaunches; ++t) h->first_start[t] = h->first_go[t] = ~0ull;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(chain<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(chain<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipStream_t st;
    (void)hipStreamCreate(&st);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float best = 1e9f;
    Ctl *res = (Ctl *)malloc(sizeof(Ctl));
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemcpy(c, h, sizeof(Ctl), hipMemcpyHostToDevice);
        (void)hipMemset(tables, 0, sizeof(unsigned long long) * kTable * 3);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a, st);
        for (int t = 0; t < kLaunches; ++t) {
            if (mode == 0)
                hipLaunchKernelGGL(chain<0>, dim3(256), dim3(768), 51 * 1024, st, c, t, work, tables, sink);
            else
                hipExtLaunchKernelGGL(chain<1>, dim3(256), dim3(768), 51 * 1024, st, nullptr, nullptr, flags, c, t, work, tables,
                                      sink);
        }
        (void)hipEventRecord(b, st);
        hipError_t e = hipEventSynchronize(b);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, a, b);
        if (e != hipSuccess) printf("error %s\n", hipGetErrorString(e));
        best = ms < best ? ms : best;
        (void)hipMemcpy(res, c, sizeof(Ctl), hipMemcpyDeviceToHost);
    }
    // how early did launch t+1 start relative to launch t's end (ticks of 10 ns), and is the workgroup -> XCD map stable?
    double early = 0, gap = 0;
    int moved = 0;
    for (int t = 1; t < kLaunches; ++t) {
        early += (double)((long long)res->last_end[t - 1] - (long long)res->first_start[t]);
        gap += (double)((long long)res->first_go[t] - (long long)res->last_end[t - 1]);
        for (int w = 0; w < 256; ++w) moved += res->xcc[t][w] != res->xcc[0][w];
    }
    unsigned long long chk[4];
    (void)hipMemcpy(chk, tables + (size_t)(kLaunches % 3) * kTable, sizeof chk, hipMemcpyDeviceToHost);
    printf("mode %d flags %d work %6d: %8.2f us per launch | next launch's first workgroup starts %7.2f us before the previous "
           "one's last end, first go %6.2f us after it | timeout %u | xcc moved %d | xcc of wg 0..9:",
           mode, flags, work, best * 1e3f / kLaunches, early / (kLaunches - 1) * 0.01, gap / (kLaunches - 1) * 0.01, res->timeout,
           moved);
    for (int w = 0; w < 10; ++w) printf(" %u", res->xcc[1][w]);
    printf(" | table[0..1] = %llu %llu (expect %d)\n", chk[0], chk[1], kLaunches * 16);
    free(res);
    (void)hipFree(c);
    (void)hipFree(tables);
    (void)hipFree(sink);
}

int main() {
    for (int work : {0, 2000, 12000}) {
        run(0, work, 0);
        run(1, work, 0);                     // spin, but ordinary (barrier bit) launches: the spin never waits
        run(1, work, hipExtAnyOrderLaunch);  // overlapped
    }
    return 0;
}
