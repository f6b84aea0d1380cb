// et_kmeans_host.inl -- part of csrc/et_kmeans.hip (ONE translation unit: this file is #included there, in order, and is not
// compiled on its own): host side: grids, workspace layout, the step API (scan / begin / assign_accumulate / update / labels / predict / init_*) and et_kmeans_init_farthest.
// clang-format off: the fragment starts and ends at namespace scope of whatever the including file has open.
}

// Grid of a grid-stride kernel sized to exactly one resident wave of workgroups (CUs x workgroups
// per CU from the occupancy query): every workgroup then gets the same number of passes (+-1) and
// there is no sparsely filled last round (4096 workgroups at 5 resident per CU would leave the
// chip 80 % idle for its fourth round).
// CU count of the CURRENT device (cached per device id; a process may drive several GPUs)
static int km_cu_count(int *dev_out = nullptr) {
    static int cu_of_device[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev_out) *dev_out = dev;
    int &n_cu = cu_of_device[dev & 63];
    if (n_cu == 0) {
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    }
    return n_cu;
}

template <typename Kernel>
static int km_resident_grid(Kernel kernel, size_t lds_bytes, int64_t work_items, int threads = kKmThreads) {
    const int n_cu = km_cu_count();
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds_bytes) != hipSuccess || per_cu < 1)
        per_cu = threads > 256 ? 1 : 4;
    int64_t g = (int64_t)n_cu * per_cu;
    if (g > kKmMaxBlocks) g = kKmMaxBlocks;
    const int64_t need = ceil_div(work_items, (int64_t)threads);
    return (int)(need < 1 ? 1 : (need < g ? need : g));
}

static int cpitch_host(int d) { return (d + 1 + 3) & ~3; }
static bool km_dims_ok(int d, int K) { return d >= 1 && d <= ET_KMEANS_MAX_D && K >= 1 && K <= ET_KMEANS_MAX_CLUSTERS; }

static size_t km_plen(int d, int K) { return (size_t)d * K + K + 2; }

// workspace carve: [block partials | block keys | cand | best (N) | labels_u8 (N) | partials | C0 scratch ...]
struct KmWorkspace {
    long long *block_partials;
    unsigned long long *block_keys;
    unsigned long long *block_keys2;  // farthest-first, single GPU: the previous step's keys (read by the next step)
    unsigned char *cand;
    long long *partials;
    et_kmeans_state *state;
    float *best;
    uint8_t *labels_u8;
    unsigned *ticket;        // arrival counter of the fused reduce + update kernel
    unsigned *init_maxabs;   // farthest-first: fp32 bits of max|x| of this shard (collected by step 1)
    uint4 *init_meta;        // farthest-first: one 16-byte summary per 256 points (kmeans_init_step_kernel)
    long long *acc_lanes;    // single-GPU fit: kAccLanes copies of every total, the assignment kernel's atomics land here
    float *last;             // single-GPU fit: centroids (d*K floats) + sim_frac (int64) of the last assignment
    long long *sim_total;    // kmeans_inertia_kernel: the exact similarity sum and the non-finite count
    et_kmeans_state *chain_state[2];  // kmeans_lloyd_chain_kernel: two copies of state / centroids / totals,
    float *chain_cen[2];              // three of the 16-copy delta table (see LloydChain)
    long long *chain_tot[2];
    long long *chain_lanes[3];
    unsigned *persist_ctl;   // kmeans_lloyd_persist_kernel: {arrivals, abort flag}, a cache line of their own
    // packed copy of the points for the trace-less chained loop (kmeans_pack_kernel); nullptr when the shape has none
    PackedHeader *pk_hdr;
    unsigned *pk_xh;
    unsigned short *pk_rr;
    float4 *pk_xa;
    size_t bytes;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// shards the packed copy is kept for: the filter's shape (d = 6, 3 <= K <= 32) and enough points.  Same-box A/B of the
// bench step over shard sizes (tools/archive/ab_packed_sizes.sh, 100 iterations): 3e5 1.41 against 1.28 ms with the fp32 filter,
// 5e5 1.60 / 1.46, 1e6 1.85 / 1.70, 2e6 2.19 / 2.19, 4e6 2.77 / 2.90, 1e7 4.40 / 5.05 -- the packed body's longer set-up
// (label table, accumulator copies) costs ~1.4 us per launch, the bytes it saves only count once a launch streams for longer
// (same-box sweep after the pass was rebuilt in round 5, bench step of 100 iterations, packed against fp32 filter: 1.0e6 points
// 2.14 / 2.06 ms, 1.5e6 2.33 / 2.43, 2.0e6 2.51 / 2.64 -- profiles/r05m)
// (... and again at the round's end -- two delta-table copies, the half-wave last drain, no private segment: 1.5e5 points 1.44 /
// 1.53 ms, 2e5 1.45 / 1.54, 4e5 1.55 / 1.62, 1e6 1.86 / 1.95; 7e4 and 1e5 level, 4e4 1.24 / 1.26 -- profiles/r05m item 14)
constexpr int64_t kPackedMinPoints = 131072;  // 2^17
static int64_t km_packed_min_points() {  // option kmeans_packed_min: tests run the packed path on small shards
    const int64_t v = options().kmeans_packed_min.load(std::memory_order_relaxed);
    return v >= 1024 ? v : kPackedMinPoints;
}
static bool km_packed_shape(int64_t N, int d, int K) { return d == 6 && K >= 3 && K <= 32 && N >= km_packed_min_points() && N % 4 == 0; }

static KmWorkspace km_carve(void *base, int64_t N, int d, int K) {
    KmWorkspace w;
    size_t off = 0;
    unsigned char *p = (unsigned char *)base;
    w.block_partials = (long long *)(p + off);
    off = align_up(off + sizeof(long long) * km_plen(d, K) * kKmMaxBlocks, 256);
    w.block_keys = (unsigned long long *)(p + off);
    off = align_up(off + sizeof(unsigned long long) * kKmMaxBlocks, 256);
    w.block_keys2 = (unsigned long long *)(p + off);
    off = align_up(off + sizeof(unsigned long long) * kKmMaxBlocks, 256);
    w.cand = p + off;
    off = align_up(off + 8 + sizeof(float) * ET_KMEANS_MAX_D, 256);
    w.partials = (long long *)(p + off);
    off = align_up(off + sizeof(long long) * km_plen(d, K), 256);
    w.state = (et_kmeans_state *)(p + off);
    off = align_up(off + sizeof(et_kmeans_state), 256);
    w.best = (float *)(p + off);
    off = align_up(off + sizeof(float) * (size_t)(N > 0 ? N : 1), 256);
    w.labels_u8 = (uint8_t *)(p + off);
    off = align_up(off + (size_t)(N > 0 ? N : 1) + 4, 256);
    w.ticket = (unsigned *)(p + off);
    off = align_up(off + sizeof(unsigned), 256);
    w.init_maxabs = (unsigned *)(p + off);
    off = align_up(off + sizeof(unsigned), 256);
    w.init_meta = (uint4 *)(p + off);
    off = align_up(off + sizeof(uint4) * (size_t)((N > 0 ? N : 1) / 256 + 2), 256);
    w.acc_lanes = (long long *)(p + off);
    off = align_up(off + sizeof(long long) * km_plen(d, K) * 16, 256);
    w.last = (float *)(p + off);
    off = align_up(off + sizeof(float) * (((size_t)d * K + 1) & ~(size_t)1) + sizeof(long long), 256);
    w.sim_total = (long long *)(p + off);
    off = align_up(off + 3 * sizeof(long long), 256);  // (sum, non-finite count, "the last launch made the sum")
    for (int i = 0; i < 2; ++i) {
        w.chain_state[i] = (et_kmeans_state *)(p + off);
        off = align_up(off + sizeof(et_kmeans_state), 256);
        w.chain_cen[i] = (float *)(p + off);
        off = align_up(off + sizeof(float) * (size_t)d * K, 256);
        w.chain_tot[i] = (long long *)(p + off);
        off = align_up(off + sizeof(long long) * km_plen(d, K), 256);
    }
    for (int i = 0; i < 3; ++i) {
        w.chain_lanes[i] = (long long *)(p + off);
        off = align_up(off + sizeof(long long) * km_plen(d, K) * 16, 256);
    }
    w.persist_ctl = (unsigned *)(p + off);
    off = align_up(off + 2 * sizeof(unsigned), 256);
    w.pk_hdr = nullptr;
    w.pk_xh = nullptr;
    w.pk_rr = nullptr;
    w.pk_xa = nullptr;
    if (km_packed_shape(N, d, K)) {  // 46 B per point
        w.pk_hdr = (PackedHeader *)(p + off);
        off = align_up(off + sizeof(PackedHeader), 256);
        w.pk_xh = (unsigned *)(p + off);
        off = align_up(off + 12 * (size_t)N, 256);
        w.pk_rr = (unsigned short *)(p + off);
        off = align_up(off + 2 * (size_t)N, 256);
        w.pk_xa = (float4 *)(p + off);
        off = align_up(off + 8192 * (((size_t)N + 255) / 256), 256);  // (whole blocks of 256 points: xa_index)
    }
    w.bytes = off;
    return w;
}

template <int D>
static int launch_assign(const float *X, int64_t N, int d, int K, const et_kmeans_state *state, const float *cen,
                         const int64_t *given, uint8_t *labels, long long *block_partials, bool vec4, hipStream_t st) {
    const size_t plen = km_plen(d, K);
    const size_t lds = sizeof(long long) * ((plen + 1) & ~(size_t)1) + sizeof(float) * (size_t)K * ((d + 1 + 3) & ~3);
    int grid;
    if (vec4) {
        grid = km_resident_grid(kmeans_assign_kernel<D, 4>, lds, N / 4);
        hipLaunchKernelGGL((kmeans_assign_kernel<D, 4>), dim3(grid), dim3(kKmThreads), lds, st, X, N, d, K, state, cen,
                           given, labels, block_partials);
    } else {
        grid = km_resident_grid(kmeans_assign_kernel<D, 1>, lds, N);
        hipLaunchKernelGGL((kmeans_assign_kernel<D, 1>), dim3(grid), dim3(kKmThreads), lds, st, X, N, d, K, state, cen,
                           given, labels, block_partials);
    }
    return grid;
}

}  // namespace et

using namespace et;

extern "C" size_t et_kmeans_partials_len(int d, int K) { return km_plen(d, K); }

extern "C" size_t et_kmeans_workspace_bytes(int64_t N, int d, int K) {
    if (!km_dims_ok(d, K) || N < 0) return 0;
    return km_carve(nullptr, N, d, K).bytes;
}

extern "C" int et_kmeans_scan(const float *X, int64_t N, int d, et_kmeans_state *state, et_stream_t stream) {
    if (!state || N < 0 || d < 1 || d > ET_KMEANS_MAX_D || (N > 0 && !X)) return ET_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(kmeans_state_reset_kernel, dim3(1), dim3(64), 0, st, state);  // zero; min_nz_x_bits = "+inf"
    ET_LAUNCH_CHECK();
    if (N == 0) return ET_OK;
    const int64_t scan_blocks = ceil_div(N * d / 4 + 1, (int64_t)kKmThreads);
    hipLaunchKernelGGL(kmeans_scan_kernel, dim3((unsigned)(scan_blocks < 1024 ? scan_blocks : 1024)), dim3(kKmThreads), 0, st,
                       X, N * d, state);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_kmeans_begin(et_kmeans_state *state, int64_t n_total, const float *centroids, int d, int K,
                               et_stream_t stream) {
    if (!state || !centroids || n_total < 0 || !km_dims_ok(d, K)) return ET_ERR_INVALID_ARG;
    hipLaunchKernelGGL(kmeans_begin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, n_total, centroids, d, K);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

// the switches of et_options.h (et_set_option), read per fit: same-process A/B runs and tests
static char km_argmax_mode() { return (char)options().kmeans_argmax.load(std::memory_order_relaxed); }
static bool km_packed_mode() { return options().kmeans_packed.load(std::memory_order_relaxed) != 0; }
static bool km_init_tiles_mode() { return options().kmeans_init_tiles.load(std::memory_order_relaxed) != 0; }
static bool km_pack_fused_mode() { return options().kmeans_pack_fused.load(std::memory_order_relaxed) != 0; }
static std::atomic<long long> g_packed_fits{0};  // fits that iterated on the packed copy (tests: the path under test ran)
#ifdef ET_TEST_HOOKS  // libetamd_testhooks.so only: problems of et_kmeans_fit_batch to treat as timed out (bit mask)
static std::atomic<unsigned long long> g_test_abort_mask{0};
extern "C" void et_testhook_kmeans_abort_mask(unsigned long long mask) { g_test_abort_mask.store(mask, std::memory_order_relaxed); }
#endif

// matrix-core filter + exact certification (default; option kmeans_argmax = v disables it)
static bool km_use_filter(const float *X, int64_t N, int d, int K, const uint8_t *labels_u8) {
    const bool vec4 = (N % 4 == 0) && aligned16(X) && ((reinterpret_cast<uintptr_t>(labels_u8) & 3u) == 0);
    return km_argmax_mode() == 'f' && vec4 && d == 6 && K >= 3 && K <= 32 && N >= 1024 && N <= 0xffffffffll;
}

// 96 KB of dynamic LDS for the fat (16-wavefront) kernels: above the default 64 KB window; the attribute is per device
static int km_fat_lds_attribute() {
    static bool lds_set[64] = {};
    int dev_id = 0;
    ET_HIP_TRY(hipGetDevice(&dev_id));
    bool &lds_ok = lds_set[dev_id & 63];
    if (lds_ok) return ET_OK;
#define ET_FAT4(KERNEL)                                                                                       \
    reinterpret_cast<const void *>(KERNEL<10, true>), reinterpret_cast<const void *>(KERNEL<10, false>),           \
        reinterpret_cast<const void *>(KERNEL<16, true>), reinterpret_cast<const void *>(KERNEL<16, false>)
    const void *fat[] = {reinterpret_cast<const void *>(kmeans_assign_filter_kernel<10>),
                         reinterpret_cast<const void *>(kmeans_assign_filter_kernel<16>),
                         ET_FAT4(kmeans_lloyd_chain_kernel), ET_FAT4(kmeans_lloyd_persist_kernel),
                         reinterpret_cast<const void *>(kmeans_lloyd_chain_kernel<10, true, true>),
                         reinterpret_cast<const void *>(kmeans_lloyd_chain_kernel<10, false, true>),
                         reinterpret_cast<const void *>(kmeans_lloyd_chain_kernel<16, true, true>),
                         reinterpret_cast<const void *>(kmeans_lloyd_chain_kernel<16, false, true>)};
#undef ET_FAT4
    for (const void *f : fat) ET_HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    lds_ok = true;
    return ET_OK;
}

static size_t km_filter_lds_bytes(int d, int K, int threads) {
    const size_t plen_ = km_plen(d, K);
    return sizeof(long long) * ((plen_ + 1) & ~(size_t)1) + sizeof(float) * (size_t)K * 8 +
           sizeof(unsigned) * kFilterQueue * (size_t)(threads / 64);
}

// 12 or 16 wavefronts per CU for a shard of N points (one workgroup per CU, 256 points per wavefront pass): the
// launch ends with its slowest wavefront, a pass costs 0.73x as much with three wavefronts per SIMD as with four.
static int km_filter_threads(int64_t N) {
    {  // option kmeans_filter_threads: measurement aid (tools/archive/ab_threads_sizes.sh)
        const int t = options().kmeans_filter_threads.load(std::memory_order_relaxed);
        if (t >= 256 && t <= kFilterMaxThreads && t % 64 == 0) return t;
    }
    int dev = 0, n_cu = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        n_cu <= 0)
        n_cu = 256;
    const int64_t groups = ceil_div(N, (int64_t)256);
    const int64_t p12 = ceil_div(groups, (int64_t)n_cu * (kFilterMinThreads / 64));
    const int64_t p16 = ceil_div(groups, (int64_t)n_cu * (kFilterMaxThreads / 64));
    return (double)p12 * 0.73 < (double)p16 ? kFilterMinThreads : kFilterMaxThreads;
}

// The loops of a single-GPU fit (chained / persistent kernel) on small shards: a launch of 256 x 12 wavefronts for a
// shard that has one or two 256-point passes per CU spends its time placing wavefronts.  Same-box sweep of the bench step
// (tools/archive/ab_threads_sizes.sh, 100 Lloyd iterations, ms): chained 256 / 512 / 768 threads per workgroup at N = 2e4
// 1.01 / 1.10 / 1.19, 7e4 1.07 / 1.14 / 1.23, 1e5 1.09 / 1.16 / 1.24, 2e5 1.25 / 1.20 / 1.29, 3e5 1.37 / 1.24 / 1.27,
// 1e6 2.65 / 1.97 / 1.65 (1024: 1.69); persistent at 2e4 0.96 / 0.99 / 1.07, at 7e4 1.50 / 1.11 / 1.17.
// -> the fewest wavefronts that still give every 256-point pass a wavefront of its own twice over.
static int km_loop_threads(int64_t N, bool persistent) {
    {  // option kmeans_filter_threads: measurement aid
        const int t = options().kmeans_filter_threads.load(std::memory_order_relaxed);
        if (t >= 256 && t <= kFilterMaxThreads && t % 64 == 0) return t;
    }
    const int n_cu = km_cu_count();
    const int64_t groups = ceil_div(N, (int64_t)256);
    if (persistent) return N <= 32768 ? 256 : 512;
    for (int t = 256; t < kFilterMinThreads; t += 256)
        if ((int64_t)n_cu * (t / 64) >= 2 * groups) return t;
    return kFilterMinThreads;
}

static int assign_accumulate_impl(const float *X, int64_t N, int d, int K, et_kmeans_state *state,
                                  const float *centroids, const int64_t *given_labels, uint8_t *labels_u8,
                                  int64_t *partials, void *workspace, size_t workspace_bytes, hipStream_t st,
                                  hipEvent_t ev_begin, hipEvent_t ev_end, bool fused_update = false, float tol = 0.f,
                                  float *trace = nullptr, bool want_sim = true) {
    if (!km_dims_ok(d, K) || N < 0 || !state || !centroids || !partials || (N > 0 && (!X || !labels_u8)))
        return ET_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < et_kmeans_workspace_bytes(N, d, K)) return ET_ERR_WORKSPACE;
    const KmWorkspace w = km_carve(workspace, N, d, K);
    const bool vec4 = (N % 4 == 0) && aligned16(X) && ((reinterpret_cast<uintptr_t>(labels_u8) & 3u) == 0);
    // The filter kernel itself runs the plain exact scan for the first iteration of a fit (state->iter == 0: no
    // labels to confirm yet).
    const bool use_filter = !given_labels && km_use_filter(X, N, d, K, labels_u8);
    int grid = 1;
    if (ev_begin) ET_HIP_TRY(hipEventRecord(ev_begin, st));
    if (use_filter) {
        int rc_attr = km_fat_lds_attribute();
        if (rc_attr) return rc_attr;
        const int threads = km_filter_threads(N);
        const size_t lds = km_filter_lds_bytes(d, K, threads);
        if (K <= 20) {
            grid = km_resident_grid(kmeans_assign_filter_kernel<10>, lds, N / 4, threads);
            hipLaunchKernelGGL(kmeans_assign_filter_kernel<10>, dim3(grid), dim3(threads), lds, st, X, N, K, state,
                               centroids, labels_u8, w.block_partials, (long long *)nullptr);
        } else {
            grid = km_resident_grid(kmeans_assign_filter_kernel<16>, lds, N / 4, threads);
            hipLaunchKernelGGL(kmeans_assign_filter_kernel<16>, dim3(grid), dim3(threads), lds, st, X, N, K, state,
                               centroids, labels_u8, w.block_partials, (long long *)nullptr);
        }
    } else if (N > 0) {
        grid = d == 6 ? launch_assign<6>(X, N, d, K, state, centroids, given_labels, labels_u8, w.block_partials, vec4, st)
                      : launch_assign<0>(X, N, d, K, state, centroids, given_labels, labels_u8, w.block_partials, vec4, st);
    } else {
        grid = launch_assign<0>(X, N, d, K, state, centroids, given_labels, labels_u8, w.block_partials, false, st);
    }
    ET_LAUNCH_CHECK();
    if (ev_end) ET_HIP_TRY(hipEventRecord(ev_end, st));
    const int plen = (int)km_plen(d, K);
    if (fused_update) {
        const size_t lds = sizeof(float) * 2 * (size_t)d * K;
        if (lds > 48 * 1024)
            ET_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kmeans_reduce_update_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kmeans_reduce_update_kernel, dim3((plen + kKmThreads / 64 - 1) / (kKmThreads / 64)),
                           dim3(kKmThreads), lds, st, w.block_partials, grid, plen, state, (long long *)partials, w.ticket, d,
                           K, tol, const_cast<float *>(centroids), trace, w.last);
    } else {
        hipLaunchKernelGGL(kmeans_reduce_partials_kernel, dim3(plen), dim3(kKmThreads), 0, st, w.block_partials, grid,
                           plen, given_labels ? 1 : 0, state, w.partials, (long long *)partials);
    }
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_kmeans_assign_accumulate(const float *X, int64_t N, int d, int K, const et_kmeans_state *state,
                                           const float *centroids, const int64_t *given_labels, uint8_t *labels_u8,
                                           int64_t *partials, void *workspace, size_t workspace_bytes,
                                           et_stream_t stream) {
    return assign_accumulate_impl(X, N, d, K, const_cast<et_kmeans_state *>(state), centroids, given_labels, labels_u8,
                                  partials, workspace, workspace_bytes, (hipStream_t)stream, nullptr, nullptr);
}

extern "C" int et_kmeans_update(et_kmeans_state *state, const int64_t *partials, int d, int K, float tol,
                                float *centroids, float *trace, et_stream_t stream) {
    if (!state || !partials || !centroids || !km_dims_ok(d, K)) return ET_ERR_INVALID_ARG;
    const size_t lds = sizeof(float) * 2 * (size_t)d * K;
    if (lds > 48 * 1024)
        ET_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kmeans_update_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kmeans_update_kernel, dim3(1), dim3(kKmThreads), lds, (hipStream_t)stream, state,
                       (const long long *)partials, d, K, tol, centroids, trace);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_kmeans_joint_done(et_kmeans_state *const *states, int n_problems, float tol, et_stream_t stream) {
    if (!states || n_problems < 1) return ET_ERR_INVALID_ARG;
    hipLaunchKernelGGL(kmeans_joint_done_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, states, n_problems, tol);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_kmeans_labels_i64(const uint8_t *labels_u8, int64_t N, int64_t *labels, et_stream_t stream) {
    if (N < 0 || (N > 0 && (!labels_u8 || !labels))) return ET_ERR_INVALID_ARG;
    if (N == 0) return ET_OK;
    hipLaunchKernelGGL(kmeans_labels_i64_kernel, dim3(km_grid(N / 4 + 1)), dim3(kKmThreads), 0, (hipStream_t)stream,
                       labels_u8, N, labels);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_kmeans_predict_batch(const float *X, int64_t x_stride, int64_t batch, int64_t N, int d,
                                       const float *centroids, int K, int64_t *labels, float *maxsims, et_stream_t stream) {
    if (!km_dims_ok(d, K) || N < 0 || batch < 0 || batch > 65535 || x_stride < 0 || !centroids || (batch * N > 0 && !X))
        return ET_ERR_INVALID_ARG;
    if (batch * N == 0) return ET_OK;
    const size_t lds = sizeof(float) * (size_t)K * ((d + 1 + 3) & ~3);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)km_grid(N), (unsigned)batch);
    if (d == 6)
        hipLaunchKernelGGL((kmeans_predict_kernel<6>), grid, dim3(kKmThreads), lds, st, X, N, d, centroids, K, labels, maxsims,
                           x_stride);
    else
        hipLaunchKernelGGL((kmeans_predict_kernel<0>), grid, dim3(kKmThreads), lds, st, X, N, d, centroids, K, labels, maxsims,
                           x_stride);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_kmeans_predict(const float *X, int64_t N, int d, const float *centroids, int K, int64_t *labels,
                                 float *maxsims, et_stream_t stream) {
    return et_kmeans_predict_batch(X, 0, 1, N, d, centroids, K, labels, maxsims, stream);
}

// fused != nullptr: single-GPU path, the one-workgroup pick launch also stores the candidate as centroid i of `fused`
// (= C0).  (Letting the last of the 4096 step workgroups do the pick was measured 3x SLOWER: 4096 device-scope
// arrivals on one ticket serialise at ~25 ns each.)
static int init_step_grid(int64_t N, int i) {
    return i == 1 ? min(km_grid(N), 1024) : min(km_grid(N / 4 + 1), 1024);  // steps >= 2: four points per lane
}

// fused != nullptr: single-GPU path.  Step i (>= 2) derives centroid i-1 itself from the keys step i-1 left in the other
// key buffer, so no pick launch separates two steps; `last` adds the pick that stores centroid i of the final step.
static int init_step_impl(const float *X, int64_t N, int d, int K, int i, const float *C0, float *best, int64_t index_base,
                          void *cand, void *workspace, size_t workspace_bytes, et_stream_t stream, float *fused,
                          bool last = true) {
    if (!km_dims_ok(d, K) || N < 0 || i < 1 || i >= K || !C0 || !cand || index_base < 0 ||
        index_base + N > 0xffffffffll || (N > 0 && (!X || !best)))
        return ET_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < et_kmeans_workspace_bytes(N, d, K)) return ET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const KmWorkspace w = km_carve(workspace, N, d, K);
    const int grid = init_step_grid(N, i);
    if (i == 1 && !fused) ET_HIP_TRY(hipMemsetAsync(w.init_maxabs, 0, sizeof(unsigned), st));  // (fused: kmeans_init_first_kernel did)
    // key buffers alternate in the fused path (a step reads its predecessor's keys while it writes its own)
    unsigned long long *keys = (fused && (i & 1)) ? w.block_keys2 : w.block_keys;
    const unsigned long long *prev = (fused && i > 1) ? ((i & 1) ? w.block_keys : w.block_keys2) : nullptr;
    const int n_prev = i > 1 ? init_step_grid(N, i - 1) : 0;
    // tile summaries: written by step 2 and every later one, used from step 3 on (K <= 32: the label set is a 32-bit mask)
    uint4 *meta = (K <= 32 && km_init_tiles_mode()) ? w.init_meta : nullptr;
    if (d == 6)
        hipLaunchKernelGGL((kmeans_init_step_kernel<6>), dim3(grid), dim3(kKmThreads), 0, st, X, N, d, K, i, C0, best,
                           w.labels_u8, w.init_maxabs, index_base, keys, prev, n_prev, fused, (unsigned char *)cand, meta,
                           i > 2 ? 1 : 0);
    else
        hipLaunchKernelGGL((kmeans_init_step_kernel<0>), dim3(grid), dim3(kKmThreads), 0, st, X, N, d, K, i, C0, best,
                           w.labels_u8, w.init_maxabs, index_base, keys, prev, n_prev, fused, (unsigned char *)cand, meta,
                           i > 2 ? 1 : 0);
    ET_LAUNCH_CHECK();
    if (!fused || last) {
        hipLaunchKernelGGL(kmeans_init_pick_kernel, dim3(1), dim3(kKmThreads), 0, st, X, N, d, keys, grid, index_base,
                           (unsigned char *)cand, fused, K, i);
        ET_LAUNCH_CHECK();
    }
    return ET_OK;
}

extern "C" int et_kmeans_init_step(const float *X, int64_t N, int d, int K, int i, const float *C0, float *best,
                                   int64_t index_base, void *cand, void *workspace, size_t workspace_bytes,
                                   et_stream_t stream) {
    return init_step_impl(X, N, d, K, i, C0, best, index_base, cand, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int et_kmeans_init_select(const void *cands, int n_cands, int stride_bytes, int d, int K, int col, float *C0,
                                     et_stream_t stream) {
    if (!cands || !C0 || !km_dims_ok(d, K) || col < 0 || col >= K || n_cands < 1 || stride_bytes < 8 + 4 * d ||
        (stride_bytes & 7) || (reinterpret_cast<uintptr_t>(cands) & 7u))
        return ET_ERR_INVALID_ARG;
    hipLaunchKernelGGL(kmeans_init_select_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                       (const unsigned char *)cands, n_cands, stride_bytes, d, K, col, C0);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_kmeans_init_set(float *C0, int d, int K, int col, const float *point, et_stream_t stream) {
    if (!C0 || !point || !km_dims_ok(d, K) || col < 0 || col >= K) return ET_ERR_INVALID_ARG;
    hipLaunchKernelGGL(kmeans_init_set_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, C0, d, K, col, point);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_kmeans_gather_point(const float *X, int64_t N, int d, int64_t local_index, float *point,
                                      et_stream_t stream) {
    if (!X || !point || d < 1 || d > ET_KMEANS_MAX_D || local_index < 0 || local_index >= N) return ET_ERR_INVALID_ARG;
    hipLaunchKernelGGL(kmeans_gather_point_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, X, N, d, local_index,
                       point);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_kmeans_init_farthest(const float *X, int64_t N, int d, int K, int64_t first_index, float *C0,
                                       void *workspace, size_t workspace_bytes, et_stream_t stream) {
    if (!km_dims_ok(d, K) || N < 1 || !X || !C0 || first_index < 0 || first_index >= N || N > 0xffffffffll)
        return ET_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < et_kmeans_workspace_bytes(N, d, K)) return ET_ERR_WORKSPACE;
    const KmWorkspace w = km_carve(workspace, N, d, K);
    float *pt = reinterpret_cast<float *>(w.cand + 8);
    hipLaunchKernelGGL(kmeans_init_first_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, X, N, d, K, first_index, C0, pt,
                       w.init_maxabs);
    ET_LAUNCH_CHECK();
    int rc = ET_OK;
    // (a one-launch form of steps 2 .. K-1 with a fence-free grid barrier was built in round 4 and lost -- 0.51 against 0.25 ms at
    // 1e7 points, no gain at 1e5: profiles/r04b_init_persist.txt; its source: tools/archive/lost_forms/kmeans_init_persist.hip.txt)
    for (int i = 1; i < K && !rc; ++i)  // one launch per new centroid (+ one pick for the last)
        rc = init_step_impl(X, N, d, K, i, C0, w.best, 0, w.cand, workspace, workspace_bytes, stream, C0, i == K - 1);
    return rc;
}
