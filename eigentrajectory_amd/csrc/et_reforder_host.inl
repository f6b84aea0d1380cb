// et_reforder_host.inl -- part of csrc/et_kmeans_reforder.hip (ONE translation unit: this file is #included there, in order, and is
// not compiled on its own): host side: workspace queries, the generic loop, the sharded run, the C ABI entry points.
using namespace et::reforder;

static size_t fast_workspace_bytes(int64_t N, int K, int64_t batch) {
    const fast::Geo g = fast::make_geo(N);
    return fast::shared_bytes(K, batch) + (size_t)batch * fast::make_layout(g, K).bytes;
}

extern "C" size_t et_kmeans_reforder_workspace_bytes(int64_t N, int d, int K) {
    if (!dims_ok(d, K) || N < 0) return 0;
    const size_t generic = carve(nullptr, N, d, K).bytes;
    const size_t quick = fast::fast_shape(N, d, K) ? fast_workspace_bytes(N, K, 1) : 0;
    return generic > quick ? generic : quick;
}

extern "C" size_t et_kmeans_reforder_batch_workspace_bytes(int64_t N, int d, int K, int64_t batch) {
    if (!dims_ok(d, K) || N < 0 || batch < 1) return 0;
    if (batch == 1) return et_kmeans_reforder_workspace_bytes(N, d, K);
    if (!fast::fast_shape(N, d, K) || batch > fast::kFMaxBatch) return 0;
    return fast_workspace_bytes(N, K, batch);
}

// the fast form (see namespace fast): all `batch` problems in one loop of one launch per iteration, joint stop
static int fast_fit(const float *X, int64_t x_stride, int64_t N, int K, int64_t batch, int max_iter, float tol, float *centroids,
                    int64_t *labels, float *trace, et_kmeans_state *states_host, et_kmeans_timing *timing_host, void *workspace,
                    hipStream_t st) {
    using namespace fast;
    Args a;
    a.geo = make_geo(N);
    a.lay = make_layout(a.geo, K);
    unsigned char *base = (unsigned char *)workspace;
    a.batch_arrive = (unsigned *)base;
    a.sq_all = (float *)(base + 256);
    a.ws = base + shared_bytes(K, batch);
    a.ws_stride = (int64_t)a.lay.bytes;
    a.X = X;
    a.x_stride = x_stride;
    a.K = K;
    a.batch = (int)batch;
    a.tol = tol;
    a.trace = trace;
    a.max_iter = max_iter;
    a.mail = nullptr;
    int rc = ET_OK;
    et::StateRing *ring = et::StateRing::get(&rc);
    if (!ring) return rc;
    a.mail = ring->mailbox_device();
    if (a.mail) ring->mailbox_reset();
    a.tiles_per_round = fast_tiles_per_round(a.geo);
    const size_t lds = fast_lds_bytes(a.geo, K, a.tiles_per_round);
    size_t ulds = 0;
    const int rows_cap = update_rows_cap(a.geo, K, (int)batch, &ulds);
    {
        static bool lds_set[64] = {};
        int dev_id = 0;
        ET_HIP_TRY(hipGetDevice(&dev_id));
        if (!lds_set[dev_id & 63]) {
            for (const void *f : {reinterpret_cast<const void *>(reforder_groups_kernel<0>), reinterpret_cast<const void *>(reforder_groups_kernel<10>),
                                  reinterpret_cast<const void *>(reforder_groups_kernel<16>)})
                ET_HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
            for (const void *f : {reinterpret_cast<const void *>(reforder_update_kernel2<kUThreads, false>),
                                  reinterpret_cast<const void *>(reforder_update_kernel2<1024, true>)})
                ET_HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kUMaxLds));
            lds_set[dev_id & 63] = true;
        }
    }
    for (int64_t b = 0; b < batch; ++b) {
        rc = et_kmeans_scan(X + b * x_stride, N, kD, (et_kmeans_state *)(a.ws + b * a.ws_stride + a.lay.state), (et_stream_t)st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(reforder_fast_prepare_kernel, dim3((unsigned)batch), dim3(kThreads), 0, st, a, (const float *)centroids);
    {
        const int64_t quads = a.geo.G << (2 * a.geo.lp);
        const int pg = (int)std::min<int64_t>((quads + kThreads - 1) / kThreads, 2048);
        hipLaunchKernelGGL(reforder_permute_kernel, dim3(pg, (unsigned)batch), dim3(kThreads), 0, st, X, x_stride, a.ws,
                           a.ws_stride, a.lay.XT, a.geo);
    }
    ET_LAUNCH_CHECK();
    hipEvent_t ev[2] = {nullptr, nullptr};
    if (timing_host) {
        ET_HIP_TRY(hipEventCreate(&ev[0]));
        ET_HIP_TRY(hipEventCreate(&ev[1]));
        ET_HIP_TRY(hipEventRecord(ev[0], st));
    }
    // the matrix-core label filter pays where the exact scan is what a launch waits for: L >= 32 (N > 4.2e6)
    const bool use_filter = a.geo.lp >= fast_filter_min_lp() && K >= 3;
    constexpr int kAhead = 16, kEvery = 4;
    et_kmeans_state *state0 = (et_kmeans_state *)(a.ws + a.lay.state);
    int launched = 0;
    bool done = false;
    const dim3 grid((unsigned)(a.geo.G + 1), (unsigned)batch), ugrid((unsigned)a.geo.n_blk, (unsigned)batch);
    // few blocks (N <= 131 072 at K <= 20): one 1024-thread workgroup per problem folds them side by side (no arrival hop)
    const int uslot = (kD * K + kFMaxK / 4 + 63) / 64 * 64;
    const bool single_update = a.geo.n_blk <= 1024 / uslot && et::options().reforder_single_update.load(std::memory_order_relaxed) != 0;
    for (int it = 0; it < max_iter && !done; ++it) {
        if (!use_filter) hipLaunchKernelGGL(reforder_groups_kernel<0>, grid, dim3(kFThreads), lds, st, a);
        else if (K <= 20) hipLaunchKernelGGL(reforder_groups_kernel<10>, grid, dim3(kFThreads), lds, st, a);
        else hipLaunchKernelGGL(reforder_groups_kernel<16>, grid, dim3(kFThreads), lds, st, a);
        if (single_update)
            hipLaunchKernelGGL((reforder_update_kernel2<1024, true>), dim3(1, (unsigned)batch), dim3(1024), ulds, st, a, rows_cap, uslot);
        else
            hipLaunchKernelGGL((reforder_update_kernel2<kUThreads, false>), ugrid, dim3(kUThreads), ulds, st, a, rows_cap, 0);
        ET_LAUNCH_CHECK();
        launched = it + 1;
        if (a.mail) {  // stay at most kAhead launches ahead of the device's report; stop when it carries the flag
            for (unsigned spins = 0;; ++spins) {
                if (ring->mailbox_done()) {
                    done = true;
                    break;
                }
                if ((long long)launched - ring->mailbox_iter() <= kAhead) break;
                if ((spins & 0xfffu) == 0xfffu && hipStreamQuery(st) == hipSuccess) break;
                sched_yield();
            }
        } else {
            if (launched % kEvery == 0) {
                rc = ring->post(state0, st, &done);
                if (rc) return rc;
            }
            ring->poll(&done);
        }
    }
    if (timing_host) ET_HIP_TRY(hipEventRecord(ev[1], st));
    const int64_t fgrid = std::min<int64_t>((N + kThreads - 1) / kThreads, 2048);
    hipLaunchKernelGGL(reforder_fast_finish_kernel, dim3((unsigned)fgrid, (unsigned)batch), dim3(kThreads), 0, st, a, centroids,
                       labels);
    ET_LAUNCH_CHECK();
    for (int64_t b = 0; b < batch; ++b)
        ET_HIP_TRY(hipMemcpyAsync(&states_host[b], a.ws + b * a.ws_stride + a.lay.state, sizeof(et_kmeans_state),
                                  hipMemcpyDeviceToHost, st));
    ET_HIP_TRY(hipStreamSynchronize(st));
    if (timing_host) {
        float ms = 0.f;
        ET_HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[1]));
        timing_host->assign_ms = ms;
        timing_host->assign_launches = launched;
        timing_host->first_assign_ms = 0.0;
        timing_host->iterations = states_host[0].iter;
        (void)hipEventDestroy(ev[0]);
        (void)hipEventDestroy(ev[1]);
    }
    for (int64_t b = 0; b < batch; ++b)
        if (states_host[b].bad_input) return ET_ERR_BAD_DATA;
    return ET_OK;
}

// ---- shards (see "The reference-order iteration over SHARDS" above) ----
namespace {
struct ShardPlan {
    int P = 0, rank = 0, tail_rank = 0, tail_full = 0, max_rows = 0, lp = 0;
    int64_t N_total = 0;
    int rows[ET_REFORDER_MAX_RANKS] = {};
    fast::Geo geo;
    fast::ShardRec rec;
    size_t off_send = 0, off_table = 0, off_rows = 0, bytes = 0;
};
int shard_plan(const int64_t *n_locals, int P, int rank, int K, ShardPlan *p) {
    using namespace fast;
    if (!n_locals || P < 1 || P > ET_REFORDER_MAX_RANKS || rank < 0 || rank >= P || K < 1 || K > kFMaxK) return ET_ERR_INVALID_ARG;
    int64_t total = 0;
    int tail_rank = 0;
    for (int r = 0; r < P; ++r) {
        if (n_locals[r] < 0) return ET_ERR_INVALID_ARG;
        total += n_locals[r];
        if (n_locals[r] > 0) tail_rank = r;
    }
    if (!fast_shape(total, kD, K)) return ET_ERR_UNSUPPORTED;
    p->lp = level_power(total / 4);
    const int64_t block = (int64_t)4 << (3 * p->lp);
    p->P = P;
    p->rank = rank;
    p->tail_rank = tail_rank;
    p->N_total = total;
    p->max_rows = 1;
    for (int r = 0; r < P; ++r) {
        if (r != tail_rank && n_locals[r] % block != 0) return ET_ERR_INVALID_ARG;  // whole level-2 blocks before the tail rank
        const Geo g = make_geo(n_locals[r], p->lp);
        p->rows[r] = g.n_blk;
        if (r == tail_rank) p->tail_full = g.full_blk;
        if (g.n_blk > p->max_rows) p->max_rows = g.n_blk;
    }
    p->geo = make_geo(n_locals[rank], p->lp);
    p->rec.max_rows = p->max_rows;
    p->rec.dk = kD * K;
    p->rec.rowlen = kD * K + kFMaxK / 4;
    size_t off = shared_bytes(K, 1) + make_layout(p->geo, K).bytes;
    p->off_send = off;
    off = up(off + sizeof(float4) * (size_t)p->rec.words());
    p->off_table = off;
    off = up(off + sizeof(float4) * (size_t)p->rec.words() * P);
    p->off_rows = off;
    off = up(off + sizeof(int) * ET_REFORDER_MAX_RANKS);
    p->bytes = off;
    return ET_OK;
}
}  // namespace

extern "C" int64_t et_kmeans_reforder_shard_block(int64_t N_total, int d, int K) {
    if (!fast::fast_shape(N_total, d, K)) return 0;
    return (int64_t)4 << (3 * level_power(N_total / 4));
}

extern "C" size_t et_kmeans_reforder_sharded_workspace_bytes(const int64_t *n_locals, int nranks, int rank, int d, int K) {
    ShardPlan p;
    if (d != fast::kD || shard_plan(n_locals, nranks, rank, K, &p) != ET_OK) return 0;
    return p.bytes;
}

// `gather(ctx, send, recv, bytes, stream)`: every rank's `bytes` at send -> recv[rank * bytes ...] on every rank (in stream
// order); `agree(ctx, state, stream)`: MAX over ranks of state->max_abs_x / bad_input.  Both nullptr: one rank.
extern "C" int et_internal_kmeans_reforder_sharded_run(const float *X, const int64_t *n_locals, int nranks, int rank, int K,
                                                       int max_iter, float tol, float *centroids, int64_t *labels, float *trace,
                                                       et_kmeans_state *state_host, void *workspace, size_t workspace_bytes,
                                                       int (*gather)(void *, const void *, void *, size_t, hipStream_t),
                                                       int (*agree)(void *, et_kmeans_state *, hipStream_t), void *ctx,
                                                       et_stream_t stream) {
    using namespace fast;
    ShardPlan p;
    int rc = shard_plan(n_locals, nranks, rank, K, &p);
    if (rc) return rc;
    if (!centroids || !state_host || !workspace || max_iter < 1 || (p.geo.N > 0 && !X)) return ET_ERR_INVALID_ARG;
    if (nranks > 1 && !gather) return ET_ERR_INVALID_ARG;
    if (workspace_bytes < p.bytes) return ET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Args a;
    a.geo = p.geo;
    a.lay = make_layout(a.geo, K);
    unsigned char *base = (unsigned char *)workspace;
    a.batch_arrive = (unsigned *)base;
    a.sq_all = (float *)(base + 256);
    a.ws = base + shared_bytes(K, 1);
    a.ws_stride = (int64_t)a.lay.bytes;
    a.X = X;
    a.x_stride = 0;
    a.K = K;
    a.batch = 1;
    a.tol = tol;
    a.trace = trace;
    a.max_iter = max_iter;
    a.mail = nullptr;
    a.tiles_per_round = fast_tiles_per_round(a.geo);
    float4 *send = (float4 *)(base + p.off_send), *table = (float4 *)(base + p.off_table);
    int *rows_dev = (int *)(base + p.off_rows);
    et::StateRing *ring = et::StateRing::get(&rc);
    if (!ring) return rc;
    const size_t lds = fast_lds_bytes(a.geo, K, a.tiles_per_round);
    size_t l2lds = 0;
    const int l2cap = update_rows_cap(a.geo, K, 1, &l2lds);
    const size_t rowb = sizeof(float4) * (size_t)p.rec.rowlen;
    const int fcap = (int)std::min<size_t>((size_t)p.max_rows, kUMaxLds / rowb);
    const size_t flds = std::max<size_t>((size_t)fcap * rowb, sizeof(float) * (size_t)kD * K);
    {
        static bool lds_set[64] = {};
        int dev_id = 0;
        ET_HIP_TRY(hipGetDevice(&dev_id));
        if (!lds_set[dev_id & 63]) {
            for (const void *f : {reinterpret_cast<const void *>(reforder_groups_kernel<0>), reinterpret_cast<const void *>(reforder_groups_kernel<10>),
                                  reinterpret_cast<const void *>(reforder_groups_kernel<16>)})
                ET_HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
            for (const void *f : {reinterpret_cast<const void *>(reforder_level2_sharded_kernel), reinterpret_cast<const void *>(reforder_finish_sharded_kernel)})
                ET_HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kUMaxLds));
            lds_set[dev_id & 63] = true;
        }
    }
    et_kmeans_state *state = (et_kmeans_state *)(a.ws + a.lay.state);
    rc = et_kmeans_scan(X, a.geo.N, kD, state, stream);
    if (rc) return rc;
    if (agree) {
        rc = agree(ctx, state, st);
        if (rc) return rc;
    }
    ET_HIP_TRY(hipMemcpyAsync(rows_dev, p.rows, sizeof(int) * (size_t)p.P, hipMemcpyHostToDevice, st));  // (p outlives the copy: this call ends with a synchronize)
    ET_HIP_TRY(hipMemsetAsync(send, 0, sizeof(float4) * (size_t)p.rec.words(), st));
    hipLaunchKernelGGL(reforder_fast_prepare_kernel, dim3(1), dim3(kThreads), 0, st, a, (const float *)centroids);
    if (a.geo.G > 0) {
        const int64_t quads = a.geo.G << (2 * a.geo.lp);
        const int pg = (int)std::min<int64_t>((quads + kThreads - 1) / kThreads, 2048);
        hipLaunchKernelGGL(reforder_permute_kernel, dim3(pg, 1), dim3(kThreads), 0, st, X, (int64_t)0, a.ws, a.ws_stride, a.lay.XT,
                           a.geo);
    }
    ET_LAUNCH_CHECK();
    const bool use_filter = a.geo.lp >= fast_filter_min_lp() && K >= 3;
    constexpr int kEvery = 4;
    bool done = false;
    const dim3 grid((unsigned)(a.geo.G + 1), 1), l2grid((unsigned)std::max(p.rows[rank], 1), 1);
    const size_t rec_bytes = sizeof(float4) * (size_t)p.rec.words();
    for (int it = 0; it < max_iter && !done; ++it) {
        if (!use_filter) hipLaunchKernelGGL(reforder_groups_kernel<0>, grid, dim3(kFThreads), lds, st, a);
        else if (K <= 20) hipLaunchKernelGGL(reforder_groups_kernel<10>, grid, dim3(kFThreads), lds, st, a);
        else hipLaunchKernelGGL(reforder_groups_kernel<16>, grid, dim3(kFThreads), lds, st, a);
        hipLaunchKernelGGL(reforder_level2_sharded_kernel, l2grid, dim3(kUThreads), l2lds, st, a, p.rec, p.rows[rank], send, l2cap);
        ET_LAUNCH_CHECK();
        if (gather) {
            rc = gather(ctx, send, table, rec_bytes, st);
            if (rc) return rc;
        } else {
            ET_HIP_TRY(hipMemcpyAsync(table, send, rec_bytes, hipMemcpyDeviceToDevice, st));
        }
        hipLaunchKernelGGL(reforder_finish_sharded_kernel, dim3(1), dim3(kUThreads), flds, st, a, p.rec, p.P, (const int *)rows_dev,
                           p.tail_rank, p.tail_full, p.N_total, (const float4 *)table, fcap);
        ET_LAUNCH_CHECK();
        // the stop flag is read one post late, by a blocking wait on that specific copy: which copy a rank sees must not
        // depend on timing, or the ranks would stop enqueueing collectives at different iterations (et_sharded.hip)
        if ((it + 1) % kEvery == 0) {
            rc = ring->post(state, st, &done);
            if (!rc && ring->pending() > 1) rc = ring->wait_oldest(&done);
            if (rc) return rc;
        }
    }
    const int64_t fgrid = std::max<int64_t>(1, std::min<int64_t>((a.geo.N + kThreads - 1) / kThreads, 2048));
    hipLaunchKernelGGL(reforder_fast_finish_kernel, dim3((unsigned)fgrid, 1), dim3(kThreads), 0, st, a, centroids,
                       a.geo.N > 0 ? labels : nullptr);
    ET_LAUNCH_CHECK();
    ET_HIP_TRY(hipMemcpyAsync(state_host, state, sizeof(et_kmeans_state), hipMemcpyDeviceToHost, st));
    ET_HIP_TRY(hipStreamSynchronize(st));
    state_host->n_total = p.N_total;
    return state_host->bad_input ? ET_ERR_BAD_DATA : ET_OK;
}

extern "C" int et_euc_sim_reforder(const float *a, const float *b, int d, int64_t m, int64_t n, float *y,
                                   et_stream_t stream) {
    if (d < 1 || d > ET_KMEANS_MAX_D || m < 0 || n < 0 || ((m > 0 && n > 0) && (!a || !b || !y))) return ET_ERR_INVALID_ARG;
    if (m == 0 || n == 0) return ET_OK;
    hipLaunchKernelGGL(reforder_euc_sim_kernel, dim3(grid_for(m * n)), dim3(kThreads), 0, (hipStream_t)stream, a, b, d, m, n, y);
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_kmeans_init_farthest_reforder(const float *X, int64_t N, int d, int K, int64_t first_index, float *C0,
                                                void *workspace, size_t workspace_bytes, et_stream_t stream) {
    if (!dims_ok(d, K) || N < 1 || !X || !C0 || first_index < 0 || first_index >= N) return ET_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < et_kmeans_reforder_workspace_bytes(N, d, K)) return ET_ERR_WORKSPACE;
    const Workspace w = carve(workspace, N, d, K);
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for(N);
    hipLaunchKernelGGL(reforder_init_pick_kernel, dim3(1), dim3(kThreads), 0, st, X, N, d, K, 0, (const Cand *)w.cands, 0,
                       first_index, C0);
    const bool incremental = d < 8 && K <= 32;  // (see reforder_init_step_inc_kernel)
    unsigned *max_abs_bits = reinterpret_cast<unsigned *>(w.counts);  // (free until a fit uses the workspace)
    const int skip_ok = N >= et::options().reforder_init_skip_min.load(std::memory_order_relaxed) ? 1 : 0;
    if (incremental) ET_HIP_TRY(hipMemsetAsync(max_abs_bits, 0, sizeof(unsigned), st));
    for (int i = 1; i < K; ++i) {
        const size_t lds = sizeof(float) * ((size_t)d * i + (size_t)i);
        // (incremental form: step i reads the candidates step i - 1 wrote -- two buffers, a late workgroup of this launch must
        // not see this launch's records -- and picks centroid i - 1 itself; only the last centroid needs the pick launch)
        Cand *mine = w.cands + (size_t)(i & 1) * kMaxBlocks;
        const Cand *prev = i > 1 ? w.cands + (size_t)((i - 1) & 1) * kMaxBlocks : nullptr;
        if (incremental && d == 6)
            hipLaunchKernelGGL(reforder_init_step_inc_kernel<6>, dim3(grid), dim3(kThreads), 0, st, X, N, d, K, i, (const float *)C0,
                               w.maxsims, w.best4, w.labels_u8, max_abs_bits, skip_ok, mine, prev, grid, C0);
        else if (incremental)
            hipLaunchKernelGGL(reforder_init_step_inc_kernel<0>, dim3(grid), dim3(kThreads), 0, st, X, N, d, K, i, (const float *)C0,
                               w.maxsims, w.best4, w.labels_u8, max_abs_bits, skip_ok, mine, prev, grid, C0);
        else
            hipLaunchKernelGGL(reforder_init_step_kernel, dim3(grid), dim3(kThreads), lds, st, X, N, d, K, i, (const float *)C0,
                               w.cands);
        if (!incremental || i == K - 1)
            hipLaunchKernelGGL(reforder_init_pick_kernel, dim3(1), dim3(kThreads), 0, st, X, N, d, K, i,
                               (const Cand *)(incremental ? mine : w.cands), grid, (int64_t)0, C0);
    }
    ET_LAUNCH_CHECK();
    return ET_OK;
}

extern "C" int et_kmeans_predict_reforder(const float *X, int64_t N, int d, const float *centroids, int K, int64_t *labels,
                                          float *maxsims, void *workspace, size_t workspace_bytes, et_stream_t stream) {
    if (!dims_ok(d, K) || N < 0 || !centroids || (N > 0 && !X)) return ET_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < et_kmeans_reforder_workspace_bytes(N, d, K)) return ET_ERR_WORKSPACE;
    if (N == 0) return ET_OK;
    const Workspace w = carve(workspace, N, d, K);
    hipStream_t st = (hipStream_t)stream;
    ET_HIP_TRY(hipMemsetAsync(w.counts, 0, sizeof(unsigned long long) * 256, st));
    hipLaunchKernelGGL(reforder_assign_kernel, dim3(grid_for(N)), dim3(kThreads), sizeof(float) * ((size_t)d * K + (size_t)K), st,
                       X, N, d, K, centroids, w.labels_u8, maxsims ? maxsims : w.maxsims, w.counts);
    ET_LAUNCH_CHECK();
    return labels ? et_kmeans_labels_i64(w.labels_u8, N, labels, stream) : ET_OK;
}

extern "C" int et_kmeans_fit_reforder(const float *X, int64_t N, int d, int K, int max_iter, float tol, float *centroids,
                                      int64_t *labels, float *trace, et_kmeans_state *state_host, void *workspace,
                                      size_t workspace_bytes, et_stream_t stream) {
    if (!dims_ok(d, K) || N < 1 || !X || !centroids || !state_host || max_iter < 1) return ET_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < et_kmeans_reforder_workspace_bytes(N, d, K)) return ET_ERR_WORKSPACE;
    if (fast::fast_shape(N, d, K))
        return fast_fit(X, 0, N, K, 1, max_iter, tol, centroids, labels, trace, state_host, nullptr, workspace, (hipStream_t)stream);
    const Workspace w = carve(workspace, N, d, K);
    hipStream_t st = (hipStream_t)stream;
    // non-finite input: reported like et_kmeans_fit does (the reference would propagate NaN)
    int rc = et_kmeans_scan(X, N, d, w.state, stream);
    if (rc) return rc;
    ET_HIP_TRY(hipMemsetAsync(w.counts, 0, sizeof(unsigned long long) * 256, st));
    ET_HIP_TRY(hipMemcpyAsync(state_host, w.state, sizeof(et_kmeans_state), hipMemcpyDeviceToHost, st));
    ET_HIP_TRY(hipStreamSynchronize(st));
    if (state_host->bad_input) return ET_ERR_BAD_DATA;
    const int lp = level_power(N / 4);
    const int64_t L = (int64_t)1 << lp;
    const int64_t full_chunks = N / 4 / L;
    const int64_t n_groups = (full_chunks + L - 1) / L;
    const size_t dk = (size_t)d * K;
    const size_t lds_assign = sizeof(float) * (dk + (size_t)K), lds_update = sizeof(float) * dk;
    const int grid = grid_for(N);
    for (int it = 0; it < max_iter; ++it) {
        hipLaunchKernelGGL(reforder_assign_kernel, dim3(grid), dim3(kThreads), lds_assign, st, X, N, d, K,
                           (const float *)centroids, w.labels_u8, w.maxsims, w.counts);
        if (n_groups > 0)
            hipLaunchKernelGGL(reforder_group_kernel, dim3(grid_for(n_groups * 4 * (int64_t)dk)), dim3(kThreads), 0, st, X, N, d, K,
                               (const uint8_t *)w.labels_u8, lp, n_groups, full_chunks, w.S1);
        hipLaunchKernelGGL(reforder_finish_kernel, dim3(1), dim3(kThreads), 0, st, X, N, d, K, (const uint8_t *)w.labels_u8, lp,
                           full_chunks, (const float *)w.S1, w.lanes, w.sums);
        hipLaunchKernelGGL(reforder_inertia_kernel, dim3(grid), dim3(kThreads), 0, st, (const float *)w.maxsims, N, w.partial);
        hipLaunchKernelGGL(reforder_update_kernel, dim3(1), dim3(kThreads), lds_update, st, w.state, (const float *)w.sums,
                           w.counts, (const double *)w.partial, grid, N, d, K, tol, centroids, trace);
        ET_LAUNCH_CHECK();
        // the reference tests `error <= tol` on the host every iteration (kmeans.py:239); so does this mode
        ET_HIP_TRY(hipMemcpyAsync(state_host, w.state, sizeof(et_kmeans_state), hipMemcpyDeviceToHost, st));
        ET_HIP_TRY(hipStreamSynchronize(st));
        if (state_host->done) break;
    }
    if (labels) {
        rc = et_kmeans_labels_i64(w.labels_u8, N, labels, stream);
        if (rc) return rc;
        ET_HIP_TRY(hipStreamSynchronize(st));
    }
    return ET_OK;
}

/* kmeans.py:228-240 for `batch` problems in ONE loop, stopped TOGETHER on the error summed over the whole (l, d, K) tensor in
 * ATen's order; d = 6, K <= 32, 1024 <= N < 2^29, batch <= 64 (batch = 1: any shape, like et_kmeans_fit_reforder). */
extern "C" int et_kmeans_fit_reforder_batch(const float *X, int64_t x_stride, int64_t N, int d, int K, int64_t batch,
                                            int max_iter, float tol, float *centroids, int64_t *labels, float *trace,
                                            et_kmeans_state *states_host, et_kmeans_timing *timing_host, void *workspace,
                                            size_t workspace_bytes, et_stream_t stream) {
    if (!dims_ok(d, K) || N < 1 || batch < 1 || !X || !centroids || !states_host || max_iter < 1) return ET_ERR_INVALID_ARG;
    const size_t need = et_kmeans_reforder_batch_workspace_bytes(N, d, K, batch);
    if (need == 0) return ET_ERR_INVALID_ARG;  // a batch of a shape the fast form does not take
    if (!workspace || workspace_bytes < need) return ET_ERR_WORKSPACE;
    if (fast::fast_shape(N, d, K))
        return fast_fit(X, x_stride, N, K, batch, max_iter, tol, centroids, labels, trace, states_host, timing_host, workspace,
                        (hipStream_t)stream);
    return et_kmeans_fit_reforder(X, N, d, K, max_iter, tol, centroids, labels, trace, states_host, workspace, workspace_bytes,
                                  stream);
}
