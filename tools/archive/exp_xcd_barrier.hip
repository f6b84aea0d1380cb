// Experiment: what does a grid barrier among 32 workgroups cost when they (a) are spread over the 8 XCDs and synchronise
// through device-scope (sc1) atomics / loads -- what kmeans_lloyd_persist_kernel does -- and (b) all sit on ONE XCD (only
// workgroups with blockIdx % 8 == 0 of a 256-workgroup launch take part) and synchronise through that XCD's L2 with
// workgroup-scope read-modify-write atomics?  Also: is the workgroup -> XCD map blockIdx % 8, launch after launch?
//   hipcc --offload-arch=gfx950 -O3 tools/exp_xcd_barrier.hip -o /tmp/exp_xcd_barrier && /tmp/exp_xcd_barrier
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int kRounds = 100, kTable = 142;

// a read that is an atomic read-modify-write (or with 0, value returned): performed where atomics are performed, never served
// by the CU's L1 (the compiler turns fetch_add(p, 0) into a plain load)
__device__ __forceinline__ unsigned rmw_read(unsigned *p) {
    unsigned r, z = 0u;
    asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(z) : "memory");
    return r;
}
__device__ __forceinline__ unsigned long long rmw_read64(unsigned long long *p) {
    unsigned long long r, z = 0ull;
    asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(z) : "memory");
    return r;
}

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }  // HW_REG_XCC_ID[3:0]

template <bool ONE_XCD, bool RMW>
__global__ __launch_bounds__(768) void barrier_kernel(unsigned *ctr, unsigned long long *table, unsigned *xcc_of, unsigned long long *ticks,
                                                      unsigned *timeouts) {
    if (threadIdx.x == 0) xcc_of[blockIdx.x] = xcc_id();
    if (ONE_XCD && (blockIdx.x & 7)) return;
    const unsigned n_wg = ONE_XCD ? gridDim.x / 8 : gridDim.x;
    const unsigned wg = ONE_XCD ? blockIdx.x / 8 : blockIdx.x;
    unsigned long long t0 = 0, sum = 0;
    if (threadIdx.x == 0) t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 1; r <= kRounds; ++r) {
        // every workgroup adds onto a small table (the Lloyd kernel's deltas), arrives, waits for all, reads the table
        if (threadIdx.x < kTable) {
            if (ONE_XCD) __hip_atomic_fetch_add(&table[threadIdx.x], (unsigned long long)(wg + r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(&table[threadIdx.x], (unsigned long long)(wg + r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) {
            if (ONE_XCD) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)r * n_wg;
            const unsigned long long s0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                const unsigned seen = RMW ? rmw_read(ctr) : __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (seen >= want) break;
                if (__builtin_amdgcn_s_memrealtime() - s0 > 1000000ull ||  // 10 ms
                    __hip_atomic_load(timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    atomicAdd(timeouts, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (__hip_atomic_load(timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;  // (uniform enough: everybody leaves soon)
        if (threadIdx.x < kTable) {
            sum += RMW ? rmw_read64(&table[threadIdx.x]) : __hip_atomic_load(&table[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) ticks[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
    if (sum == 12345ull) ticks[0] = 0;
}

template <bool ONE_XCD, bool RMW>
static void run(int n_wg_launch, const char *what) {
    unsigned *ctr, *xcc, *timeouts;
    unsigned long long *table, *ticks;
    (void)hipMalloc(&ctr, 256);
    (void)hipMalloc(&timeouts, 256);
    (void)hipMalloc(&xcc, 4 * 256);
    (void)hipMalloc(&table, 8 * kTable);
    (void)hipMalloc(&ticks, 8 * 256);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipMemset(ctr, 0, 256);
        (void)hipMemset(timeouts, 0, 256);
        (void)hipMemset(table, 0, 8 * kTable);
        (void)hipMemset(ticks, 0, 8 * 256);
        hipLaunchKernelGGL((barrier_kernel<ONE_XCD, RMW>), dim3(n_wg_launch), dim3(768), 0, 0, ctr, table, xcc, ticks, timeouts);
        (void)hipDeviceSynchronize();
        std::vector<unsigned> hx(256);
        std::vector<unsigned long long> ht(256), tb(kTable);
        unsigned to = 0;
        (void)hipMemcpy(hx.data(), xcc, 4 * n_wg_launch, hipMemcpyDeviceToHost);
        (void)hipMemcpy(ht.data(), ticks, 8 * n_wg_launch, hipMemcpyDeviceToHost);
        (void)hipMemcpy(tb.data(), table, 8 * kTable, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost);
        int map_ok = 1;
        for (int b = 0; b < n_wg_launch; ++b) map_ok &= hx[b] == hx[b % 8] && (b < 8 || true);
        unsigned mask = 0;
        for (int b = 0; b < n_wg_launch; b += ONE_XCD ? 8 : 1) mask |= 1u << hx[b];
        const int n_wg = ONE_XCD ? n_wg_launch / 8 : n_wg_launch;
        unsigned long long expect = 0;
        for (int r = 1; r <= kRounds; ++r)
            for (int w = 0; w < n_wg; ++w) expect += (unsigned long long)(w + r);
        printf("%-46s %3d workgroups: %6.2f us per round | xcc of blocks 0..7: %u %u %u %u %u %u %u %u, map b %% 8 stable: %d, xcc set of the "
               "participants 0x%02x | time-outs %u | table %s\n",
               what, n_wg, (double)ht[0] * 0.01 / kRounds, hx[0], hx[1], hx[2], hx[3], hx[4], hx[5], hx[6], hx[7], map_ok, mask, to,
               tb[0] == expect ? "ok" : "WRONG");
        fflush(stdout);
    }
    (void)hipFree(ctr);
    (void)hipFree(timeouts);
    (void)hipFree(xcc);
    (void)hipFree(table);
    (void)hipFree(ticks);
}

int main() {
    run<false, false>(32, "8 XCDs, reads = device-scope loads");
    run<false, true>(32, "8 XCDs, reads = atomic or 0");
    run<true, false>(256, "one XCD, reads = device-scope loads");
    run<true, true>(256, "one XCD, reads = atomic or 0");
    run<true, true>(128, "one XCD, reads = atomic or 0");
    return 0;
}
