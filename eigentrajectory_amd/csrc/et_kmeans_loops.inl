// et_kmeans_loops.inl -- part of csrc/et_kmeans.hip (ONE translation unit: this file is #included there, in order, and is not
// compiled on its own): host side: the chained loop, the persistent loop, et_kmeans_fit and et_kmeans_fit_batch.
// clang-format off: the fragment starts and ends at namespace scope of whatever the including file has open.
// Everything the chained loop's buffers need before its first launch, in ONE launch (the six separate copies / fills
// it replaces were ~5 us packets each): state and centroids into copy 0, totals of copy 0 and the three delta tables zeroed.
__global__ __launch_bounds__(kKmThreads) void kmeans_chain_prepare_kernel(const et_kmeans_state *__restrict__ state,
                                                                          const float *__restrict__ cen, int dk, int plen,
                                                                          LloydChain first, long long *lanes_b, long long *lanes_c) {
    const int tid = blockIdx.x * kKmThreads + threadIdx.x, n_thr = gridDim.x * kKmThreads;
    constexpr int kStateWords = (int)(sizeof(et_kmeans_state) / sizeof(unsigned));
    if (tid < kStateWords) reinterpret_cast<unsigned *>(first.st_wr)[tid] = reinterpret_cast<const unsigned *>(state)[tid];
    for (int e = tid; e < dk; e += n_thr) first.cen_wr[e] = cen[e];
    for (int e = tid; e < plen; e += n_thr) first.tot_wr[e] = 0;
    for (int e = tid; e < plen * kAccLanes; e += n_thr) {
        first.lanes_wr[e] = 0;
        lanes_b[e] = 0;
        lanes_c[e] = 0;
    }
}

// A collective the chained loop runs between two launches when the points are sharded over ranks: SUM over ranks of
// `count` int64 values, in place, enqueued on `st` (csrc/et_sharded.hip binds it to ncclAllReduce).  Without one
// (single GPU) the loop also polls the convergence flag opportunistically; with one, every rank must enqueue the same
// collectives, so the flag is read by a blocking wait on a specific, long-arrived copy (et_hostring.h).
struct ChainHook {
    int (*reduce)(void *ctx, long long *buf, size_t count, hipStream_t st) = nullptr;
    void *ctx = nullptr;
};

// The chained Lloyd loop (kmeans_lloyd_chain_kernel): `state` holds the initial state block (after scan / begin) on
// entry and the final one on return, `centroids` the initial / final centroids, `partials` receives the final totals.
// Ends with the update of the last assignment (finalize kernel) and, for trace-less fits, the inertia pass.
static int km_chain_run(const float *X, int64_t N, int d, int K, int max_iter, float tol, float *centroids,
                        uint8_t *labels_u8, float *trace, et_kmeans_state *state, long long *partials, const KmWorkspace &w,
                        hipStream_t st, ChainHook hook, std::vector<hipEvent_t> *events, int time_every, int *launched_out) {
    constexpr int kEvery = 4;
    auto timed = [&](int it) { return events && (it == 0 || it % time_every == 1); };  // (time_every >= kTimedRun)
    int rc = ET_OK;
    StateRing *ring = StateRing::get(&rc);
    if (!ring) return rc;
    const bool want_sim = trace != nullptr;
    const int threads = km_loop_threads(N, false);
    const size_t plen = km_plen(d, K), lds = km_filter_lds_bytes(d, K, threads);
    // rows that allow 16-byte loads and enough points: the filter body; any other shard of a sharded fit: the exact scan,
    // one point per lane, inside the same kernel (a single-GPU fit only comes here with vec_ok)
    const bool vec_ok = km_use_filter(X, N, d, K, labels_u8);
    const int64_t work_items = vec_ok ? N / 4 : N;
    rc = km_fat_lds_attribute();
    if (rc) return rc;
    // single GPU: the kernel reports (done, iterations applied) into the ring's pinned mailbox and nothing is copied
    // inside the loop; sharded: the lockstep state copies below (what a rank reads from a mailbox depends on timing)
    unsigned long long *mail = hook.reduce ? nullptr : ring->mailbox_device();
    if (mail) ring->mailbox_reset();
    constexpr int kAhead = 16;  // launches the host may be ahead of the device's report
    {
        LloydChain first{};  // the kernel writes through the *_wr fields: copy 0 of state / centroids / totals, table 0
        first.st_wr = w.chain_state[0];
        first.cen_wr = w.chain_cen[0];
        first.tot_wr = w.chain_tot[0];
        first.lanes_wr = w.chain_lanes[0];
        hipLaunchKernelGGL(kmeans_chain_prepare_kernel, dim3(8), dim3(kKmThreads), 0, st, (const et_kmeans_state *)state,
                           (const float *)centroids, d * K, (int)plen, first, w.chain_lanes[1], w.chain_lanes[2]);
        ET_LAUNCH_CHECK();
    }
    // trace-less fits of big shards iterate on the packed copy (ET_KMEANS_PACKED=0: the fp32 filter, for A/B runs)
    const bool packed = vec_ok && !want_sim && w.pk_xh && km_packed_mode();
    const bool pack_fused = km_pack_fused_mode();
    if (packed) g_packed_fits.fetch_add(1, std::memory_order_relaxed);
    if (packed && !pack_fused) {
        const int64_t quads = N / 4;
        const int pgrid = (int)std::min<int64_t>((quads + kKmThreads - 1) / kKmThreads, 1024);
        hipLaunchKernelGGL(kmeans_pack_kernel, dim3(pgrid), dim3(kKmThreads), 0, st, X, N, (const et_kmeans_state *)state,
                           w.pk_hdr, w.pk_xh, w.pk_rr, w.pk_xa);
        ET_LAUNCH_CHECK();
    }
    int chain_copies = options().kmeans_chain_copies.load(std::memory_order_relaxed);
    if (chain_copies != 1 && chain_copies != 2 && chain_copies != 4 && chain_copies != 8) chain_copies = 2;
    while (chain_copies > 1 && (size_t)chain_copies * compact_pitch((int)plen) > plen * kAccLanes) chain_copies >>= 1;
    auto chain_for = [&](int t) {
        LloydChain ch;
        ch.st_rd = w.chain_state[t & 1];
        ch.st_wr = w.chain_state[(t + 1) & 1];
        ch.cen_rd = w.chain_cen[t & 1];
        ch.cen_wr = w.chain_cen[(t + 1) & 1];
        ch.tot_rd = w.chain_tot[t & 1];
        ch.tot_wr = w.chain_tot[(t + 1) & 1];
        ch.lanes_rd = w.chain_lanes[t % 3];
        ch.lanes_wr = w.chain_lanes[(t + 1) % 3];
        ch.lanes_zero = w.chain_lanes[(t + 2) % 3];
        ch.last = w.last;
        ch.mail = mail;
        // ONE compact copy of the delta table in every form of the chained loop: measured against the 16-copy table on one
        // GPU it is 1.0 us per launch FASTER (50.0 against 51.0 us, three alternating runs: the prologue reads 142 values
        // instead of folding 2272, and <= 256 arrivals per address spread over the launch's tail are absorbed by the
        // memory side), and it is what a sharded fit puts on the wire
        ch.compact = 1;
        // ... and on one GPU a few of them (option kmeans_chain_copies, default 2): a launch's 256 workgroups add their deltas
        // within a few microseconds of each other, and arrivals on one address are served one after the other
        ch.copies = hook.reduce ? 1 : chain_copies;
        ch.vec_ok = vec_ok ? 1 : 0;
        ch.pk = packed ? LloydPacked{w.pk_xh, w.pk_rr, w.pk_xa, w.pk_hdr, pack_fused ? 1 : 0} : LloydPacked{nullptr, nullptr, nullptr, nullptr, 0};
        return ch;
    };
    int grid = 0, launched = 0;
    bool done = false;
    // A trace-less fit evaluates the inertia of its last assignment in a pass of its own over the points (48 us at 1e7
    // points + two packets).  When the loop runs to max_iter its last launch is known beforehand: that one launch takes the
    // form that accumulates the exact similarity sum (the traced fits' kernel on the fp32 rows, +9 us at 1e7 points), the
    // finalize kernel's update turns the sum into the inertia -- the same integers, the same bits -- and the pass is skipped
    // ON THE DEVICE (sim_total[2]): a fit that converges earlier never reaches that launch's assignment and keeps the pass.
    // (the decision must not depend on this rank's shard: a shard without 16-byte rows runs the exact scan, which adds the
    // similarity sum in every launch -- so every rank of a sharded fit feeds the last launch's sum and skips the pass)
    const bool sim_tail = !want_sim && max_iter >= 2;
    bool last_was_sim = false;
    for (int it = 0; it < max_iter && !done; ++it) {
        const LloydChain ch = chain_for(it);
        const bool sim_now = want_sim || (sim_tail && it == max_iter - 1);
        last_was_sim = sim_now && !want_sim;
        if (timed(it)) ET_HIP_TRY(hipEventRecord((*events)[2 * it], st));
#define ET_LAUNCH_CHAIN(NR, SIM)                                                                                          \
    do {                                                                                                                  \
        if (!grid) {                                                                                                      \
            grid = km_resident_grid(kmeans_lloyd_chain_kernel<NR, SIM>, lds, work_items, threads);                        \
            const int cap = options().kmeans_loop_grid.load(std::memory_order_relaxed);                                   \
            if (cap > 0 && grid > cap) grid = cap;                                                                        \
        }                                                                                                                 \
        if (it == 0)                                                                                                      \
            hipLaunchKernelGGL((kmeans_lloyd_chain_kernel<NR, SIM, true>), dim3(grid), dim3(threads), lds, st, X, N, K,   \
                               ch, labels_u8, tol, trace, 0);                                                             \
        else                                                                                                              \
            hipLaunchKernelGGL((kmeans_lloyd_chain_kernel<NR, SIM, false>), dim3(grid), dim3(threads), lds, st, X, N, K,  \
                               ch, labels_u8, tol, trace, 1);                                                             \
    } while (0)
        if (K <= 20) {
            if (sim_now) ET_LAUNCH_CHAIN(10, true);
            else ET_LAUNCH_CHAIN(10, false);
        } else {
            if (sim_now) ET_LAUNCH_CHAIN(16, true);
            else ET_LAUNCH_CHAIN(16, false);
        }
#undef ET_LAUNCH_CHAIN
        ET_LAUNCH_CHECK();
        // (the end event of a timed launch: right after the first launch -- the exact scan --, after the FOURTH launch of a
        // sampled run of the others: an event between two kernels costs a dispatch gap on each side, and a bracket around
        // one 33-us launch measured 37.5 us where rocprofv3 saw 32.8; four launches per bracket measure the period)
        if (events && it == 0) ET_HIP_TRY(hipEventRecord((*events)[1], st));
        if (events && it >= kTimedRun && timed(it - (kTimedRun - 1)) && it - (kTimedRun - 1) != 0)
            ET_HIP_TRY(hipEventRecord((*events)[2 * (it - (kTimedRun - 1)) + 1], st));
        // sharded: the deltas this launch added onto its (one-copy, compact) table become the sum over all ranks' before
        // the next launch reads them: d K + K + 2 int64, 1.1 KB for d = 6, K = 20
        if (hook.reduce) {
            rc = hook.reduce(hook.ctx, ch.lanes_wr, plen, st);
            if (rc) return rc;
        }
        launched = it + 1;
        if (mail) {
            // launch `it` reports iter = it (it applied the update of assignment it - 1); stay at most kAhead launches
            // ahead of the last report, stop as soon as a report carries the flag
            for (unsigned spins = 0;; ++spins) {
                if (ring->mailbox_done()) {
                    done = true;
                    break;
                }
                if ((long long)launched - ring->mailbox_iter() <= kAhead) break;
                // (a stream query puts a marker into the queue: only as the rare safety net against a lost report --
                // if everything launched so far has finished, what the mailbox says is final)
                if ((spins & 0xfffu) == 0xfffu && hipStreamQuery(st) == hipSuccess) break;
                sched_yield();
            }
        } else {
            if (launched % kEvery == 0) {
                rc = ring->post(ch.st_wr, st, &done);
                if (!rc && hook.reduce && ring->pending() > 1) rc = ring->wait_oldest(&done);  // the same copy on every rank
                if (rc) return rc;
            }
            if (!hook.reduce) ring->poll(&done);
        }
    }
    const LloydChain ch = chain_for(launched);
    const size_t flds = 4096 + sizeof(long long) * plen;
    hipLaunchKernelGGL(kmeans_chain_finalize_kernel, dim3(1), dim3(kKmThreads), flds, st, ch, state, partials, centroids, d,
                       K, tol, trace, launched > 0 ? 1 : 0, want_sim ? (long long *)nullptr : w.sim_total,
                       (last_was_sim && launched == max_iter) ? 1 : 0);
    ET_LAUNCH_CHECK();
    if (!want_sim) {  // inertia of the last assignment (over all ranks' points); the finalize kernel zeroed sim_total
        const size_t ilds = sizeof(float) * (size_t)K * cpitch_host(d);
        // (every workgroup ends with one device-scope atomic on the same address, ~15 ns each: 4096 workgroups made the
        // pass atomic-bound at 64 us; four points per lane and 1024 workgroups stream instead)
        const int igrid = min(km_grid(N / 4 + 1), 1024);
        hipLaunchKernelGGL((kmeans_inertia_kernel<6>), dim3(igrid), dim3(kKmThreads), ilds, st, X, N, d, K,
                           (const float *)w.last, (const uint8_t *)labels_u8, w.sim_total, (int64_t)0, (int64_t)0,
                           (const long long *)(w.sim_total + 2));
        ET_LAUNCH_CHECK();
        if (hook.reduce) {
            rc = hook.reduce(hook.ctx, w.sim_total, 2, st);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(kmeans_inertia_finish_kernel, dim3(1), dim3(64), 0, st, state, (const float *)w.last, d, K,
                           (const long long *)w.sim_total, (int64_t)0, (const long long *)(w.sim_total + 2));
        ET_LAUNCH_CHECK();
    }
    if (launched_out) *launched_out = launched;
    return ET_OK;
}

// ---- the persistent loop (kmeans_lloyd_persist_kernel) ----
// Its grid barrier needs every workgroup of the launch resident at once.  One launch alone always is (the grid is one
// resident round, km_resident_grid); several fits running side by side in this process (the ten initialisations of the
// sklearn recipe, the moving / static clusterings, BatchKMeans' problems -- one host thread and stream each) share the
// CUs through this counter: a fit takes as many CU slots as its grid has workgroups before it launches and gives them
// back after its final synchronisation, so the persistent grids in flight never need more CUs than the device has.
// (Other kernels may occupy CUs for a while -- they end; another PROCESS on the same GPU can break the promise, which
// is what the kernel's time-out and the chained fallback are for.)
#include <condition_variable>
#include <mutex>
namespace et {
class PersistSlots {
  public:
    static PersistSlots &of_device(int dev) {
        static PersistSlots slots[64];
        return slots[dev & 63];
    }
    void acquire(int n, int capacity) {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return used_ == 0 || used_ + n <= capacity; });
        used_ += n;
    }
    void release(int n) {
        {
            std::lock_guard<std::mutex> lk(m_);
            used_ -= n;
        }
        cv_.notify_all();
    }

  private:
    std::mutex m_;
    std::condition_variable cv_;
    int used_ = 0;
};

}  // namespace et
namespace et {

__global__ __launch_bounds__(kKmThreads) void kmeans_persist_prepare_kernel(int plen, long long *l0, long long *l1, long long *l2,
                                                                            unsigned *ctl, long long *sim_total,
                                                                            int64_t ws_stride) {
    {   // blockIdx.y: problem of a batch (workspaces of identical layout, ws_stride bytes apart)
        const int64_t off = (int64_t)blockIdx.y * ws_stride;
        l0 = reinterpret_cast<long long *>(reinterpret_cast<char *>(l0) + off);
        l1 = reinterpret_cast<long long *>(reinterpret_cast<char *>(l1) + off);
        l2 = reinterpret_cast<long long *>(reinterpret_cast<char *>(l2) + off);
        ctl = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(ctl) + off);
        sim_total = reinterpret_cast<long long *>(reinterpret_cast<char *>(sim_total) + off);
    }
    const int tid = blockIdx.x * kKmThreads + threadIdx.x, n_thr = gridDim.x * kKmThreads;
    for (int e = tid; e < plen * kAccLanes; e += n_thr) {
        l0[e] = 0;
        l1[e] = 0;
        l2[e] = 0;
    }
    if (tid < 2) {
        ctl[tid] = 0u;
        sim_total[tid] = 0;
    }
}

// Which loop form a single-GPU fit takes.  Measured per iteration (tools/archive/ab_loop_sizes.py, profiles/r03f_loop_sizes.txt,
// same box): N = 2e4 11.9 us persistent / 13.7 chained, 7e4 12.9 / 13.9, 1e5 13.6 / 14.0, 3e5 16.2 / 13.6,
// 1e6 20.9 / 15.4, 1e7 56.1 / 49.7 -- the grid barrier + fold + update of the persistent form (~4 us for 33 workgroups,
// ~7 us + the spread of 256 workgroups' finishing times for a full grid) beats a kernel boundary only while the grid is
// small; for a full grid the staggered start of a new launch's workgroups happens to hide the uneven pass counts that
// the barrier exposes.  Hence: persistent up to kPersistMaxPoints, chained above; option kmeans_loop = persist / chain forces one.
// (round 3, later: with 256-thread workgroups the chained loop is ahead from ~3e4 points on -- the table above
// km_loop_threads; et_kmeans_fit_batch keeps the persistent form for its side-by-side problems at any size it takes)
constexpr int64_t kPersistMaxPoints = 32768;
static char km_persist_mode() { return (char)options().kmeans_loop.load(std::memory_order_relaxed); }  // 'a'uto, 'c'hain, 'p'ersist
static bool km_persist_wanted(int64_t N) {
    const char mode = km_persist_mode();
    if (mode == 'c') return false;
    if (mode == 'p') return true;
    return N <= kPersistMaxPoints;
}

// All Lloyd iterations of a single-GPU fit in one launch.  Same contract as km_chain_run; *aborted = true (and nothing
// written to state / centroids / partials) when the grid barrier timed out -- the caller repeats the fit chained.
static int km_persist_run(const float *X, int64_t N, int d, int K, int max_iter, float tol, float *centroids,
                          uint8_t *labels_u8, float *trace, et_kmeans_state *state, long long *partials, const KmWorkspace &w,
                          hipStream_t st, hipEvent_t ev_begin, hipEvent_t ev_end, bool *aborted, int *grid_out) {
    const bool want_sim = trace != nullptr;
    const int threads = km_loop_threads(N, true);
    const size_t plen = km_plen(d, K), lds = km_filter_lds_bytes(d, K, threads);
    int rc = km_fat_lds_attribute();
    if (rc) return rc;
    hipLaunchKernelGGL(kmeans_persist_prepare_kernel, dim3(8), dim3(kKmThreads), 0, st, (int)plen, w.chain_lanes[0],
                       w.chain_lanes[1], w.chain_lanes[2], w.persist_ctl, w.sim_total, (int64_t)0);
    ET_LAUNCH_CHECK();
    LloydPersist pa;
    pa.st_in = state;
    pa.cen_in = centroids;
    pa.st_out = w.chain_state[0];  // staged: the caller's buffers are only written once the loop is known to have run
    pa.cen_out = w.chain_cen[0];
    pa.tot_out = w.chain_tot[0];
    pa.lanes0 = w.chain_lanes[0];
    pa.lanes1 = w.chain_lanes[1];
    pa.lanes2 = w.chain_lanes[2];
    pa.arrive = w.persist_ctl;
    pa.abort = w.persist_ctl + 1;
    pa.last = w.last;
    pa.ws_stride = pa.x_stride = pa.cen_stride = 0;
    int grid = 0, dev = 0;
    const int n_cu = km_cu_count(&dev);
    // one resident round of workgroups of the instantiation that is launched
    if (K <= 20) grid = want_sim ? km_resident_grid(kmeans_lloyd_persist_kernel<10, true>, lds, N / 4, threads)
                                 : km_resident_grid(kmeans_lloyd_persist_kernel<10, false>, lds, N / 4, threads);
    else grid = want_sim ? km_resident_grid(kmeans_lloyd_persist_kernel<16, true>, lds, N / 4, threads)
                         : km_resident_grid(kmeans_lloyd_persist_kernel<16, false>, lds, N / 4, threads);
    if (grid > n_cu) grid = n_cu;  // one fat workgroup per CU is what the co-residency accounting assumes
    PersistSlots &slots = PersistSlots::of_device(dev);
    slots.acquire(grid, n_cu);
    struct Release {
        PersistSlots &s;
        int n;
        ~Release() { s.release(n); }
    } release_on_exit{slots, grid};
    if (ev_begin) ET_HIP_TRY(hipEventRecord(ev_begin, st));
#define ET_LAUNCH_PERSIST(NR, SIM)                                                                                \
    hipLaunchKernelGGL((kmeans_lloyd_persist_kernel<NR, SIM>), dim3(grid), dim3(threads), lds, st, X, N, K, pa,      \
                       labels_u8, tol, trace, max_iter)
    if (K <= 20) {
        if (want_sim) ET_LAUNCH_PERSIST(10, true);
        else ET_LAUNCH_PERSIST(10, false);
    } else {
        if (want_sim) ET_LAUNCH_PERSIST(16, true);
        else ET_LAUNCH_PERSIST(16, false);
    }
#undef ET_LAUNCH_PERSIST
    ET_LAUNCH_CHECK();
    if (ev_end) ET_HIP_TRY(hipEventRecord(ev_end, st));
    if (grid_out) *grid_out = grid;
    // the one host round trip of the fit that the chained loop does not have: did the barrier hold?  (pinned staging
    // would save nothing here: the caller synchronises right after this anyway)
    unsigned ctl[2] = {0u, 0u};
    ET_HIP_TRY(hipMemcpyAsync(ctl, w.persist_ctl, sizeof ctl, hipMemcpyDeviceToHost, st));
    ET_HIP_TRY(hipStreamSynchronize(st));
    *aborted = ctl[1] != 0u;
    if (*aborted) return ET_OK;
    ET_HIP_TRY(hipMemcpyAsync(state, w.chain_state[0], sizeof(et_kmeans_state), hipMemcpyDeviceToDevice, st));
    ET_HIP_TRY(hipMemcpyAsync(centroids, w.chain_cen[0], sizeof(float) * (size_t)d * K, hipMemcpyDeviceToDevice, st));
    ET_HIP_TRY(hipMemcpyAsync(partials, w.chain_tot[0], sizeof(long long) * plen, hipMemcpyDeviceToDevice, st));
    if (!want_sim) {  // inertia of the last assignment; the prepare kernel zeroed sim_total
        const size_t ilds = sizeof(float) * (size_t)K * cpitch_host(d);
        const int igrid = min(km_grid(N / 4 + 1), 1024);
        hipLaunchKernelGGL((kmeans_inertia_kernel<6>), dim3(igrid), dim3(kKmThreads), ilds, st, X, N, d, K,
                           (const float *)w.last, (const uint8_t *)labels_u8, w.sim_total);
        hipLaunchKernelGGL(kmeans_inertia_finish_kernel, dim3(1), dim3(64), 0, st, state, (const float *)w.last, d, K,
                           (const long long *)w.sim_total);
        ET_LAUNCH_CHECK();
    }
    return ET_OK;
}
}  // namespace et

#ifdef ET_EXP_WAITSTAMP
extern "C" int et_debug_waitstamp(unsigned long long *host, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(et::g_waitstamp), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(et::g_waitstamp), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
extern "C" int et_debug_prostamp(unsigned long long *host, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(et::g_prostamp), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(et::g_prostamp), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
#endif
#ifdef ET_PERSIST_STAMPS
extern "C" int et_debug_persist_stamps(void *host, size_t bytes) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(et::g_persist_stamps), bytes) == hipSuccess ? 0 : 1;
}
#endif

// Entry points for csrc/et_sharded.hip (not part of the public header): can this rank's shard run the chained loop,
// and the loop itself with a reduction between the launches.  `workspace` as for et_kmeans_fit.
// (the choice depends on d, K and the process-wide ET_KMEANS_ARGMAX setting only -- never on a rank's own shard --, so the
// ranks of a sharded fit agree on the loop form, i.e. on the collectives they enqueue, without exchanging anything)
extern "C" long long et_internal_kmeans_packed_fits(void) { return g_packed_fits.load(std::memory_order_relaxed); }

extern "C" int et_internal_kmeans_chain_usable(int d, int K) {
    return km_dims_ok(d, K) && km_argmax_mode() == 'f' && d == 6 && K >= 3 && K <= 32 ? 1 : 0;
}
extern "C" int et_internal_kmeans_chain_run(const float *X, int64_t N, int d, int K, int max_iter, float tol, float *centroids,
                                            uint8_t *labels_u8, float *trace, et_kmeans_state *state, int64_t *partials,
                                            void *workspace, size_t workspace_bytes,
                                            int (*reduce)(void *, long long *, size_t, hipStream_t), void *ctx,
                                            et_stream_t stream) {
    if (!workspace || workspace_bytes < et_kmeans_workspace_bytes(N, d, K)) return ET_ERR_WORKSPACE;
    const KmWorkspace w = km_carve(workspace, N, d, K);
    ChainHook hook;
    hook.reduce = reduce;
    hook.ctx = ctx;
    return km_chain_run(X, N, d, K, max_iter, tol, centroids, labels_u8, trace, state, (long long *)partials, w,
                        (hipStream_t)stream, hook, nullptr, 8, nullptr);
}

// ---- several fits side by side in ONE persistent launch (blockIdx.y = problem) ----
// The reference's anchor clustering is ten independent small fits (sklearn's n_init = 10, anchor.py:65-71) on the same
// points; at dataset sizes each is latency bound and driving them from ten host threads / streams scaled to barely 2x
// (profiles/r03g_concurrent_inits.txt: 3.4 ms of stream / thread set-up, fits slowed down by each other).  Here the
// problems are the y dimension of ONE persistent grid: each has its own workspace (same layout, ws_stride apart), its
// own barrier counter and stops on its own error; problems whose workgroups do not all fit on the device at once are
// launched in chunks.
namespace et {
__global__ __launch_bounds__(kKmThreads) void kmeans_batch_collect_kernel(const float *cen_staged, const et_kmeans_state *st_staged,
                                                                          et_kmeans_state *st_final, int64_t ws_stride,
                                                                          float *centroids, int dk, const unsigned *ctl) {
    const int64_t off = (int64_t)blockIdx.x * ws_stride;
    // a problem whose grid barrier timed out never wrote its staged results: leave the caller's (initial) centroids and
    // the begun state alone -- the host repeats that fit from them with the chained loop
    if (byte_shift(ctl, off)[1] != 0u) return;
    const float *src = byte_shift(cen_staged, off);
    for (int e = threadIdx.x; e < dk; e += kKmThreads) centroids[(int64_t)blockIdx.x * dk + e] = src[e];
    constexpr int kStateWords = (int)(sizeof(et_kmeans_state) / sizeof(unsigned));
    if ((int)threadIdx.x < kStateWords)
        reinterpret_cast<unsigned *>(byte_shift(st_final, off))[threadIdx.x] =
            reinterpret_cast<const unsigned *>(byte_shift(st_staged, off))[threadIdx.x];
}
}  // namespace et

extern "C" int et_kmeans_fit(const float *X, int64_t N, int d, int K, int max_iter, float tol, float *centroids,
                             int64_t *labels, float *trace, et_kmeans_state *state_host,
                             et_kmeans_timing *timing_host, void *workspace, size_t workspace_bytes,
                             et_stream_t stream);

extern "C" size_t et_kmeans_batch_workspace_bytes(int64_t N, int d, int K, int64_t batch) {
    const size_t one = et_kmeans_workspace_bytes(N, d, K);
    return one == 0 || batch < 1 ? 0 : one * (size_t)batch;
}

extern "C" int et_kmeans_fit_batch(const float *X, int64_t x_stride, int64_t N, int d, int K, int64_t batch, int max_iter,
                                   float tol, float *centroids, int64_t *labels, et_kmeans_state *states_host,
                                   void *workspace, size_t workspace_bytes, et_stream_t stream) {
    if (!km_dims_ok(d, K) || N < 1 || !X || !centroids || !states_host || max_iter < 1 || batch < 1 || batch > 65535 ||
        x_stride < 0)
        return ET_ERR_INVALID_ARG;
    const size_t one = et_kmeans_workspace_bytes(N, d, K);
    if (!workspace || workspace_bytes < one * (size_t)batch) return ET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const KmWorkspace w = km_carve(workspace, N, d, K);  // problem 0's; problem b's is the same layout, b * one bytes on
    const int64_t dk = (int64_t)d * K;
    // every problem alone through et_kmeans_fit: shapes the persistent kernel does not take, or when it is switched off
    // (NaN / Inf in one problem does not stop the others, but is reported: ET_ERR_BAD_DATA after the loop, like the
    // side-by-side path)
    auto one_by_one = [&](int64_t from, int64_t to) -> int {
        bool bad = false;
        for (int64_t b = from; b < to; ++b) {
            const int rc = et_kmeans_fit(X + b * x_stride, N, d, K, max_iter, tol, centroids + b * dk,
                                         labels ? labels + b * N : nullptr, nullptr,
                                         &states_host[b], nullptr, byte_shift((char *)workspace, b * (int64_t)one), one, stream);
            if (rc && rc != ET_ERR_BAD_DATA) return rc;
            bad = bad || rc == ET_ERR_BAD_DATA;
        }
        return bad ? ET_ERR_BAD_DATA : ET_OK;
    };
    bool takes = km_persist_mode() != 'c' && x_stride % 4 == 0;
    for (int64_t b = 0; takes && b < batch; ++b) takes = km_use_filter(X + b * x_stride, N, d, K, w.labels_u8);
    const int threads = km_filter_threads(N);
    const size_t plen = km_plen(d, K), lds = km_filter_lds_bytes(d, K, threads);
    int dev = 0;
    const int n_cu = km_cu_count(&dev);
    int grid = 0;
    if (takes) {
        int rc = km_fat_lds_attribute();
        if (rc) return rc;
        grid = K <= 20 ? km_resident_grid(kmeans_lloyd_persist_kernel<10, false>, lds, N / 4, threads)
                       : km_resident_grid(kmeans_lloyd_persist_kernel<16, false>, lds, N / 4, threads);
        if (grid > n_cu / 2) takes = false;  // a shard that fills the device by itself: nothing to put side by side
    }
    if (!takes) {
        return one_by_one(0, batch);
    }
    const bool shared = x_stride == 0;
    // scale scan: once when the problems share their points, else per problem; then every problem's begin in one launch
    for (int64_t b = 0; b < (shared ? 1 : batch); ++b) {
        const int rc = et_kmeans_scan(X + b * x_stride, N, d, byte_shift(w.state, b * (int64_t)one), stream);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(kmeans_begin_kernel, dim3((unsigned)batch), dim3(64), 0, st, w.state, N, (const float *)centroids, d, K,
                       (int64_t)one, dk, shared ? 1 : 0);
    ET_LAUNCH_CHECK();
    const int per_launch = n_cu / grid;  // problems whose workgroups are resident together (one fat workgroup per CU)
    PersistSlots &slots = PersistSlots::of_device(dev);
    std::vector<unsigned> ctl((size_t)batch * 2, 0u);
    for (int64_t b0 = 0; b0 < batch; b0 += per_launch) {
        const int64_t nb = batch - b0 < per_launch ? batch - b0 : per_launch;
        const int64_t off = b0 * (int64_t)one;
        hipLaunchKernelGGL(kmeans_persist_prepare_kernel, dim3(8, (unsigned)nb), dim3(kKmThreads), 0, st, (int)plen,
                           byte_shift(w.chain_lanes[0], off), byte_shift(w.chain_lanes[1], off),
                           byte_shift(w.chain_lanes[2], off), byte_shift(w.persist_ctl, off), byte_shift(w.sim_total, off),
                           (int64_t)one);
        ET_LAUNCH_CHECK();
        LloydPersist pa;
        pa.st_in = byte_shift(w.state, off);
        pa.cen_in = centroids + b0 * dk;
        pa.st_out = byte_shift(w.chain_state[0], off);
        pa.cen_out = byte_shift(w.chain_cen[0], off);
        pa.tot_out = byte_shift(w.chain_tot[0], off);
        pa.lanes0 = byte_shift(w.chain_lanes[0], off);
        pa.lanes1 = byte_shift(w.chain_lanes[1], off);
        pa.lanes2 = byte_shift(w.chain_lanes[2], off);
        pa.arrive = byte_shift(w.persist_ctl, off);
        pa.abort = byte_shift(w.persist_ctl, off) + 1;
        pa.last = byte_shift(w.last, off);
        pa.ws_stride = (int64_t)one;
        pa.x_stride = x_stride;
        pa.cen_stride = dk;
        slots.acquire(grid * (int)nb, n_cu);
        struct Release {
            PersistSlots &s;
            int n;
            ~Release() { s.release(n); }
        } release_on_exit{slots, grid * (int)nb};
        const dim3 g((unsigned)grid, (unsigned)nb);
        if (K <= 20)
            hipLaunchKernelGGL((kmeans_lloyd_persist_kernel<10, false>), g, dim3(threads), lds, st, X + b0 * x_stride, N, K, pa,
                               byte_shift(w.labels_u8, off), tol, (float *)nullptr, max_iter);
        else
            hipLaunchKernelGGL((kmeans_lloyd_persist_kernel<16, false>), g, dim3(threads), lds, st, X + b0 * x_stride, N, K, pa,
                               byte_shift(w.labels_u8, off), tol, (float *)nullptr, max_iter);
        ET_LAUNCH_CHECK();
#ifdef ET_TEST_HOOKS  // libetamd_testhooks.so only (tests/test_gpu_kmeans.py): bit mask of problems to treat as timed out
        if (const unsigned long long mask = g_test_abort_mask.load(std::memory_order_relaxed)) {
            for (int64_t b = b0; b < b0 + nb; ++b)
                if (b < 64 && (mask >> b & 1ull))
                    ET_HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(byte_shift(w.persist_ctl, b * (int64_t)one) + 1), 1, 1, st));
        }
#endif
        // did every problem's barrier hold?  (the slots go back when this chunk has run)
        ET_HIP_TRY(hipMemcpy2DAsync(ctl.data() + 2 * b0, 2 * sizeof(unsigned), byte_shift(w.persist_ctl, off), one,
                                    2 * sizeof(unsigned), (size_t)nb, hipMemcpyDeviceToHost, st));
        ET_HIP_TRY(hipStreamSynchronize(st));
    }
    // results: staged centroids / state -> the caller's (B, d, K) array and the problems' state blocks; the inertia of the
    // last assignment; the labels when asked for
    hipLaunchKernelGGL(kmeans_batch_collect_kernel, dim3((unsigned)batch), dim3(kKmThreads), 0, st, (const float *)w.chain_cen[0],
                       (const et_kmeans_state *)w.chain_state[0], w.state, (int64_t)one, centroids, (int)dk,
                       (const unsigned *)w.persist_ctl);
    {
        const size_t ilds = sizeof(float) * (size_t)K * cpitch_host(d);
        const int igrid = min(km_grid(N / 4 + 1), 1024);
        hipLaunchKernelGGL((kmeans_inertia_kernel<6>), dim3(igrid, (unsigned)batch), dim3(kKmThreads), ilds, st, X, N, d, K,
                           (const float *)w.last, (const uint8_t *)w.labels_u8, w.sim_total, (int64_t)one, x_stride);
        hipLaunchKernelGGL(kmeans_inertia_finish_kernel, dim3((unsigned)batch), dim3(64), 0, st, w.state, (const float *)w.last, d,
                           K, (const long long *)w.sim_total, (int64_t)one);
    }
    if (labels)
        hipLaunchKernelGGL(kmeans_labels_i64_kernel, dim3(km_grid(N / 4 + 1), (unsigned)batch), dim3(kKmThreads), 0, st,
                           (const uint8_t *)w.labels_u8, N, labels, (int64_t)one);
    ET_LAUNCH_CHECK();
    ET_HIP_TRY(hipMemcpy2DAsync(states_host, sizeof(et_kmeans_state), w.state, one, sizeof(et_kmeans_state), (size_t)batch,
                                hipMemcpyDeviceToHost, st));
    ET_HIP_TRY(hipStreamSynchronize(st));
    // a problem whose grid barrier timed out (another process on the GPU): that fit again, alone, with the chained loop
    for (int64_t b = 0; b < batch; ++b) {
        if (ctl[2 * b + 1] == 0u) continue;
        const int rc = one_by_one(b, b + 1);
        if (rc && rc != ET_ERR_BAD_DATA) return rc;  // bad data: states_host[b].bad_input is set, reported below
    }
    for (int64_t b = 0; b < batch; ++b)
        if (states_host[b].bad_input) return ET_ERR_BAD_DATA;
    return ET_OK;
}

extern "C" int et_kmeans_fit(const float *X, int64_t N, int d, int K, int max_iter, float tol, float *centroids,
                             int64_t *labels, float *trace, et_kmeans_state *state_host,
                             et_kmeans_timing *timing_host, void *workspace, size_t workspace_bytes,
                             et_stream_t stream) {
    if (!km_dims_ok(d, K) || N < 1 || !X || !centroids || !state_host || max_iter < 1) return ET_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < et_kmeans_workspace_bytes(N, d, K)) return ET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const KmWorkspace w = km_carve(workspace, N, d, K);
    // Timing (optional): HIP events around a SAMPLE of the assign launches -- the first one (plain exact scan) and
    // every kTimeEvery-th of the others (filter kernel).  An event record between two kernels costs a ~5 us
    // dispatch gap on each side, so timing every launch would slow the loop it measures by ~15 %.
    constexpr int kTimeEvery = 8;
    auto timed = [&](int it) { return timing_host && (it == 0 || it % kTimeEvery == 1); };
    // timing events belong to the device that is current when they are created: one cached set per (host thread, device)
    int dev_id = 0;
    ET_HIP_TRY(hipGetDevice(&dev_id));
    struct EventHolder {  // destroyed with the host thread
        std::vector<std::vector<hipEvent_t>> v;
        ~EventHolder() {
            for (auto &dev_events : v)
                for (hipEvent_t e : dev_events) (void)hipEventDestroy(e);
        }
    };
    static thread_local EventHolder per_thread;
    std::vector<std::vector<hipEvent_t>> &per_device_events = per_thread.v;
    if ((int)per_device_events.size() <= dev_id) per_device_events.resize(dev_id + 1);
    std::vector<hipEvent_t> &events = per_device_events[dev_id];
    if (timing_host) {
        while ((int)events.size() < 2 * max_iter) {
            hipEvent_t e;
            ET_HIP_TRY(hipEventCreate(&e));
            events.push_back(e);
        }
    }
    // The reference synchronises every iteration (error <= tol on the host, kmeans.py:239).  Here convergence
    // lives on the device: once state->done is set the remaining launches are no-ops.  The host never waits
    // for it inside the loop (et_hostring.h): the queue stays at most kSlots * kEvery iterations ahead.
    constexpr int kEvery = 4;
    int rc = ET_OK;
    StateRing *ring = StateRing::get(&rc);
    if (!ring) return rc;
    rc = et_kmeans_scan(X, N, d, w.state, stream);
    if (!rc) rc = et_kmeans_begin(w.state, N, centroids, d, K, stream);
    if (rc) return rc;
    // Without a trace the inertia of an iteration is not an output (kmeans.py:234 only prints it and keeps the last
    // one): the assignment kernels skip the fp64 similarity sums and the inertia of the LAST assignment is evaluated
    // by one extra pass after the loop -- the same bits as the per-iteration sum would have given.
    const bool want_sim = trace != nullptr;
    int launched = 0;
    bool done = false;
    // shards the matrix-core filter takes: the chained form (kmeans_lloyd_chain_kernel) -- every launch applies the
    // previous iteration's update in its prologue, in every workgroup; one more update after the loop.  (A one-launch
    // form with a ticketed fold + update in the last workgroup served shards <= 131072 points until its serial tail
    // lost to this prologue: 19.8 against 16.8 us per iteration at N = 1e5.)
    const bool chained = km_use_filter(X, N, d, K, w.labels_u8);
    // ... as ONE persistent launch for all iterations (kmeans_lloyd_persist_kernel); one launch per iteration
    // (km_chain_run, the form the sharded loop uses) if its grid barrier timed out or ET_KMEANS_LOOP=chain asks for it
    bool persisted = false;
    if (chained && km_persist_wanted(N)) {
        bool aborted = false;
        rc = km_persist_run(X, N, d, K, max_iter, tol, centroids, w.labels_u8, trace, w.state, (long long *)w.partials, w, st,
                            timing_host ? events[0] : nullptr, timing_host ? events[1] : nullptr, &aborted, nullptr);
        if (rc) return rc;
        persisted = !aborted;
    }
    if (chained && !persisted) {
        rc = km_chain_run(X, N, d, K, max_iter, tol, centroids, w.labels_u8, trace, w.state, (long long *)w.partials, w, st,
                          ChainHook{}, timing_host ? &events : nullptr, kTimeEvery, &launched);
        if (rc) return rc;
    }
    if (!chained) {
        ET_HIP_TRY(hipMemsetAsync(w.ticket, 0, sizeof(unsigned), st));
        ET_HIP_TRY(hipMemsetAsync(w.acc_lanes, 0, sizeof(long long) * km_plen(d, K) * 16, st));
    }
    for (int it = 0; !chained && it < max_iter && !done; ++it) {
        rc = assign_accumulate_impl(X, N, d, K, w.state, centroids, nullptr, w.labels_u8, (int64_t *)w.partials, workspace,
                                    workspace_bytes, st, timed(it) ? events[2 * it] : nullptr,
                                    timed(it) ? events[2 * it + 1] : nullptr, true, tol, trace, want_sim);
        if (rc) return rc;
        launched = it + 1;
        if (launched % kEvery == 0) {
            rc = ring->post(w.state, st, &done);
            if (rc) return rc;
        }
        ring->poll(&done);
    }
    if (!want_sim && !chained) {
        ET_HIP_TRY(hipMemsetAsync(w.sim_total, 0, 2 * sizeof(long long), st));
        const size_t ilds = sizeof(float) * (size_t)K * cpitch_host(d);
        const int igrid = km_grid(N);
        if (d == 6)
            hipLaunchKernelGGL((kmeans_inertia_kernel<6>), dim3(igrid), dim3(kKmThreads), ilds, st, X, N, d, K,
                               (const float *)w.last, (const uint8_t *)w.labels_u8, w.sim_total);
        else
            hipLaunchKernelGGL((kmeans_inertia_kernel<0>), dim3(igrid), dim3(kKmThreads), ilds, st, X, N, d, K,
                               (const float *)w.last, (const uint8_t *)w.labels_u8, w.sim_total);
        hipLaunchKernelGGL(kmeans_inertia_finish_kernel, dim3(1), dim3(64), 0, st, w.state, (const float *)w.last, d, K,
                           (const long long *)w.sim_total);
        ET_LAUNCH_CHECK();
    }
    if (labels) {  // (NULL: the caller only wants the centroids)
        rc = et_kmeans_labels_i64(w.labels_u8, N, labels, stream);
        if (rc) return rc;
    }
    ET_HIP_TRY(hipMemcpyAsync(state_host, w.state, sizeof(et_kmeans_state), hipMemcpyDeviceToHost, st));
    ET_HIP_TRY(hipStreamSynchronize(st));
    if (timing_host && persisted) {
        // one launch ran every assignment (the exact first pass included) and every update of the fit
        float ms = 0.f;
        ET_HIP_TRY(hipEventElapsedTime(&ms, events[0], events[1]));
        timing_host->assign_ms = (double)ms;
        timing_host->assign_launches = 1;
        timing_host->first_assign_ms = 0.0;
        timing_host->iterations = state_host->iter;
    } else if (timing_host) {
        // launches after convergence are no-ops (a few microseconds); count only the working ones.  The first
        // launch of a fit is the plain exact scan with full accumulation, the others the filter kernel.
        const int worked = (int)(state_host->iter < launched ? state_host->iter : launched);
        double total = 0.0, first = 0.0;
        int samples = 0;
        for (int it = 0; it < worked; ++it) {
            if (!timed(it)) continue;
            if (it > 0 && it + kTimedRun - 1 >= worked) continue;  // (a run that the fit's end cut short has no end event)
            float ms = 0.f;
            ET_HIP_TRY(hipEventElapsedTime(&ms, events[2 * it], events[2 * it + 1]));
            if (it == 0) {
                first = (double)ms;
            } else {
                total += (double)ms;
                samples += kTimedRun;
            }
        }
        timing_host->assign_ms = total;
        timing_host->assign_launches = samples;
        timing_host->first_assign_ms = first;
        timing_host->iterations = samples;
    }
    return state_host->bad_input ? ET_ERR_BAD_DATA : ET_OK;
}
