// Second overlap experiment (see exp_overlap.hip): the hand-off between two overlapped launches WITHOUT cache fences.
// exp_overlap showed (profiles/r03a_overlap.txt) that launches without the barrier bit do overlap (no time-outs, correct
// sums, the workgroup -> XCD map is the same in every launch) but that system-scope fences cost 20-100 us per launch.
// Here every value that crosses launches travels through agent-scope (sc1) atomics / loads / stores, so no buffer_wbl2 /
// buffer_inv is needed at all:
//   SYNC 0: ordinary launches, plain loads (the reference point)      SYNC 1: spin + agent-scope FENCES (acquire all threads,
//   release thread 0)      SYNC 2: spin + sc1 loads of the table, s_waitcnt + barrier before the counter, no cache fence
//   FAT: 90 KB of LDS per workgroup so that two workgroups cannot share a CU (like the Lloyd kernel: 116 VGPRs x 12 wavefronts)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kLaunches = 100;
constexpr int kTable = 142 * 16;

struct Ctl {
    unsigned done[kLaunches + 4];
    unsigned timeout;
    unsigned pad[3];
    unsigned long long first_start[kLaunches], last_end[kLaunches], first_go[kLaunches];
};

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }

template <int SYNC, bool FAT>
__global__ __launch_bounds__(768) void chain(Ctl *c, int t, int work, unsigned long long *tables, float *sink,
                                             unsigned char *labels) {
    extern __shared__ unsigned char lds[];
    __shared__ unsigned long long sSum;
    unsigned long long t0 = 0;
    if (threadIdx.x == 0) {
        t0 = now();
        sSum = 0;
        atomicMin(&c->first_start[t], t0);
        if (SYNC >= 1 && t > 0) {
            while (__hip_atomic_load(&c->done[t - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
                if (now() - t0 > 5000000ull) {  // 50 ms
                    c->timeout = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        atomicMin(&c->first_go[t], now());
    }
    __syncthreads();
    if (SYNC == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const unsigned long long *rd = tables + (size_t)(t % 3) * kTable;
    unsigned long long *wr = tables + (size_t)((t + 1) % 3) * kTable;
    unsigned long long *zr = tables + (size_t)((t + 2) % 3) * kTable;
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < kTable; i += blockDim.x)
        s += SYNC == 2 ? __hip_atomic_load(&rd[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : rd[i];
    // a label word per thread, written back through to memory (what the Lloyd kernel would do with its changed labels)
    unsigned *lw = reinterpret_cast<unsigned *>(labels) + (size_t)blockIdx.x * 768 + threadIdx.x;
    const unsigned lab = SYNC == 2 ? __hip_atomic_load(lw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *lw;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sSum, s);
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < kTable; i += blockDim.x) {
            if (SYNC == 2) __hip_atomic_store(&zr[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else zr[i] = 0;
        }
    __syncthreads();
    float v = (float)threadIdx.x + (float)(sSum & 0xff);
    float keep[FAT ? 96 : 1];
#pragma unroll
    for (int j = 0; j < (FAT ? 96 : 1); ++j) keep[j] = v + (float)j;
    for (int i = 0; i < work; ++i) {
        v = fmaf(v, 1.0001f, 0.5f);
        if (FAT) keep[i % 96 == 0 ? 0 : 1] += v;  // cheap, keeps the array live
    }
    if (FAT) {
#pragma unroll
        for (int j = 0; j < 96; ++j) v += keep[j];
    }
    if (v == 12345.678f) sink[0] = v;
    lds[threadIdx.x] = (unsigned char)v;
    if (threadIdx.x < 142) atomicAdd(&wr[threadIdx.x * 16 + (blockIdx.x & 15)], (unsigned long long)(t + 1));
    if ((threadIdx.x & 31) == 0) {  // a sparse changed-label write
        if (SYNC == 2) __hip_atomic_store(lw, lab + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *lw = lab + 1u;
    }
    if (SYNC == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // s_waitcnt: this wavefront's stores / atomics are done
    __syncthreads();
    if (threadIdx.x == 0) {
        if (SYNC == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(&c->done[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicMax(&c->last_end[t], now());
    }
}

template <int SYNC, bool FAT>
static void run(int work, int flags, bool ext_api) {
    Ctl *c;
    unsigned long long *tables;
    float *sink;
    unsigned char *labels;
    (void)hipMalloc(&c, sizeof(Ctl));
    (void)hipMalloc(&tables, sizeof(unsigned long long) * kTable * 3);
    (void)hipMalloc(&sink, 64);
    (void)hipMalloc(&labels, 256 * 768 * 4);
    std::vector<unsigned char> init(sizeof(Ctl), 0);
    Ctl *h = reinterpret_cast<Ctl *>(init.data());
    for (int t = 0; t < kLaunches; ++t) h->first_start[t] = h->first_go[t] = ~0ull;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(chain<SYNC, FAT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              96 * 1024);
    hipStream_t st;
    (void)hipStreamCreate(&st);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float best = 1e9f;
    Ctl *res = (Ctl *)malloc(sizeof(Ctl));
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipMemcpy(c, h, sizeof(Ctl), hipMemcpyHostToDevice);
        (void)hipMemset(tables, 0, sizeof(unsigned long long) * kTable * 3);
        (void)hipMemset(labels, 0, 256 * 768 * 4);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a, st);
        for (int t = 0; t < kLaunches; ++t) {
            if (!ext_api)
                hipLaunchKernelGGL((chain<SYNC, FAT>), dim3(256), dim3(768), (FAT ? 90 : 51) * 1024, st, c, t, work, tables, sink, labels);
            else
                hipExtLaunchKernelGGL((chain<SYNC, FAT>), dim3(256), dim3(768), (FAT ? 90 : 51) * 1024, st, nullptr, nullptr, flags, c, t, work,
                                      tables, sink, labels);
        }
        (void)hipEventRecord(b, st);
        hipError_t e = hipEventSynchronize(b);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, a, b);
        if (e != hipSuccess) printf("error %s\n", hipGetErrorString(e));
        best = ms < best ? ms : best;
        (void)hipMemcpy(res, c, sizeof(Ctl), hipMemcpyDeviceToHost);
    }
    double early = 0, gap = 0;
    for (int t = 1; t < kLaunches; ++t) {
        early += (double)((long long)res->last_end[t - 1] - (long long)res->first_start[t]);
        gap += (double)((long long)res->first_go[t] - (long long)res->last_end[t - 1]);
    }
    unsigned long long chk[2];
    unsigned lchk[2];
    (void)hipMemcpy(chk, tables + (size_t)(kLaunches % 3) * kTable, sizeof chk, hipMemcpyDeviceToHost);
    (void)hipMemcpy(lchk, labels, sizeof lchk, hipMemcpyDeviceToHost);
    printf("sync %d fat %d api %s flags %d work %6d: %7.2f us per launch | starts %6.2f us before prev end, go %5.2f us after | "
           "timeout %u | table %llu (expect %d) label %u (expect %d)\n",
           SYNC, (int)FAT, ext_api ? "ext" : "std", flags, work, best * 1e3f / kLaunches, early / (kLaunches - 1) * 0.01,
           gap / (kLaunches - 1) * 0.01, res->timeout, chk[0], kLaunches * 16, lchk[0], kLaunches);
    fflush(stdout);
    free(res);
    (void)hipFree(c);
    (void)hipFree(tables);
    (void)hipFree(sink);
    (void)hipFree(labels);
    (void)hipStreamDestroy(st);
}

int main() {
    for (int work : {0, 3000}) {
        run<0, false>(work, 0, false);
        run<0, false>(work, 0, true);
        run<0, true>(work, 0, false);
        run<1, true>(work, 0, true);
        run<1, true>(work, hipExtAnyOrderLaunch, true);
        run<2, true>(work, 0, true);
        run<2, true>(work, hipExtAnyOrderLaunch, true);
        run<2, false>(work, hipExtAnyOrderLaunch, true);
    }
    return 0;
}
