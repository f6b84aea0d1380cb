// et_sharded.hip -- the data-sharded fit and k-means as ONE host call each, RCCL collectives on the caller's stream.
//
// One process per GPU; the N trajectories / points are split over the ranks of an ncclComm_t (RCCL over xGMI).
// Rows are independent everywhere except in two places:
//
//   et_fit_gram_sharded        local fp64 Gram matrices + row count, then ONE grouped all-reduce(SUM) of
//                              (2 T_obs)^2 + (2 T_pred)^2 doubles + 1 int64 (6.7 KB); every rank then runs the same
//                              deterministic eigensolver, so no broadcast of U is needed.
//   et_kmeans_init_farthest_sharded   per new centroid: local candidate -> all-gather of one 8 + 4 d byte record per
//                              rank -> the same arg-min on every rank.
//   et_kmeans_fit_sharded      all-reduce MAX/MIN of the scale scan once; per Lloyd iteration ONE launch of the chained
//                              kernel (csrc/et_kmeans.hip) + ONE all-reduce(SUM) of its d K + K + 2 int64 deltas (1.1 KB)
//                              in place, all enqueued on one stream with no host round trip (shapes the chained kernel
//                              does not take: assignment kernels -> the same all-reduce -> update kernel); convergence is
//                              decided on the device from identical integers on every rank and looked at from the host a
//                              few iterations late (et_hostring.h), so every rank enqueues the same collectives.
// Exact 64-bit fixed-point sums make the k-means result bit-identical for any number of ranks / any partition.
//
// RCCL is bound at run time (dlopen + dlsym): libetamd.so has no link-time dependency on it, and a process that has
// already loaded an RCCL (PyTorch ships one) gets THAT instance, not a second copy.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>

#include "et_common.h"
#include "et_hostring.h"

namespace et {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    bool ok = false;
};

static Rccl g_rccl;

// path of an RCCL that this process has already mapped (e.g. torch/lib/librccl.so), or "" -- so that we bind the
// instance the host framework uses instead of loading a second one
static void mapped_rccl_path(char *out, size_t cap) {
    out[0] = 0;
    FILE *f = fopen("/proc/self/maps", "r");
    if (!f) return;
    char line[1024];
    while (fgets(line, sizeof line, f)) {
        const char *p = strstr(line, "librccl");
        if (!p) continue;
        const char *start = strchr(line, '/');
        if (!start) continue;
        size_t len = strcspn(start, "\n");
        if (len >= cap) len = cap - 1;
        memcpy(out, start, len);
        out[len] = 0;
        break;
    }
    fclose(f);
}

static int rccl_load(const char *path) {
    if (g_rccl.ok) return ET_OK;
    void *h = nullptr;
    char mapped[1024];
    if (path && path[0]) h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        mapped_rccl_path(mapped, sizeof mapped);
        if (mapped[0]) h = dlopen(mapped, RTLD_NOW | RTLD_GLOBAL);
    }
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) return ET_ERR_RCCL;
    Rccl r;
    r.handle = h;
#define ET_SYM(field, name) \
    *reinterpret_cast<void **>(&r.field) = dlsym(h, name); \
    if (!r.field) return ET_ERR_RCCL
    ET_SYM(GetUniqueId, "ncclGetUniqueId");
    ET_SYM(CommInitRank, "ncclCommInitRank");
    ET_SYM(CommDestroy, "ncclCommDestroy");
    ET_SYM(CommCount, "ncclCommCount");
    ET_SYM(CommUserRank, "ncclCommUserRank");
    ET_SYM(AllReduce, "ncclAllReduce");
    ET_SYM(AllGather, "ncclAllGather");
    ET_SYM(GroupStart, "ncclGroupStart");
    ET_SYM(GroupEnd, "ncclGroupEnd");
#undef ET_SYM
    r.ok = true;
    g_rccl = r;
    return ET_OK;
}

#define ET_RCCL_TRY(expr)                          \
    do {                                           \
        if ((expr) != ncclSuccess) return ET_ERR_RCCL; \
    } while (0)

static size_t rec_bytes(int d) { return (size_t)((8 + 4 * d + 7) / 8 * 8); }

}  // namespace et

using namespace et;

extern "C" int et_comm_load(const char *librccl_path) { return rccl_load(librccl_path); }

extern "C" int et_comm_unique_id(void *id128) {
    if (!id128) return ET_ERR_INVALID_ARG;
    int rc = rccl_load(nullptr);
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == ET_COMM_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ET_RCCL_TRY(g_rccl.GetUniqueId(reinterpret_cast<ncclUniqueId *>(id128)));
    return ET_OK;
}

extern "C" int et_comm_init_rank(const void *id128, int nranks, int rank, et_comm_t *comm) {
    if (!id128 || !comm || nranks < 1 || rank < 0 || rank >= nranks) return ET_ERR_INVALID_ARG;
    int rc = rccl_load(nullptr);
    if (rc) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    ET_RCCL_TRY(g_rccl.CommInitRank(&c, nranks, id, rank));
    *comm = (et_comm_t)c;
    return ET_OK;
}

extern "C" int et_comm_destroy(et_comm_t comm) {
    if (!comm) return ET_OK;
    if (!g_rccl.ok) return ET_ERR_RCCL;
    ET_RCCL_TRY(g_rccl.CommDestroy((ncclComm_t)comm));
    return ET_OK;
}

extern "C" int et_comm_info(et_comm_t comm, int *nranks, int *rank) {
    if (!comm || !g_rccl.ok) return ET_ERR_INVALID_ARG;
    if (nranks) ET_RCCL_TRY(g_rccl.CommCount((ncclComm_t)comm, nranks));
    if (rank) ET_RCCL_TRY(g_rccl.CommUserRank((ncclComm_t)comm, rank));
    return ET_OK;
}

// ---------------------------------------------------------------------------------------------- fit
extern "C" int et_fit_gram_sharded(const float *obs, const float *pred, int64_t N, int T_obs, int T_pred, int mode,
                                   float static_dist, int which, double *G_obs, double *G_pred, int64_t *count,
                                   void *workspace, size_t workspace_bytes, et_comm_t comm, et_stream_t stream) {
    int rc = et_fit_gram(obs, pred, N, T_obs, T_pred, mode, static_dist, which, G_obs, G_pred, count, workspace,
                         workspace_bytes, stream);
    if (rc) return rc;
    if (!comm) return ET_OK;  // a single shard
    if (!g_rccl.ok) return ET_ERR_RCCL;
    hipStream_t st = (hipStream_t)stream;
    ncclComm_t c = (ncclComm_t)comm;
    // one grouped launch: 2 fp64 matrices + the int64 row count (exact), 6.7 KB for T = 8 / 12
    ET_RCCL_TRY(g_rccl.GroupStart());
    ET_RCCL_TRY(g_rccl.AllReduce(G_obs, G_obs, (size_t)4 * T_obs * T_obs, ncclDouble, ncclSum, c, st));
    ET_RCCL_TRY(g_rccl.AllReduce(G_pred, G_pred, (size_t)4 * T_pred * T_pred, ncclDouble, ncclSum, c, st));
    ET_RCCL_TRY(g_rccl.AllReduce(count, count, 1, ncclInt64, ncclSum, c, st));
    ET_RCCL_TRY(g_rccl.GroupEnd());
    return ET_OK;
}

// ------------------------------------------------------------------------------------------ k-means
extern "C" size_t et_kmeans_sharded_workspace_bytes(int64_t N_local, int d, int K, int nranks) {
    const size_t base = et_kmeans_workspace_bytes(N_local, d, K);
    if (base == 0 || nranks < 1) return 0;
    // + the gathered candidate records, one local record and a d-float staging point
    return base + 256 + rec_bytes(d) * (size_t)(nranks + 1) + 256 + sizeof(float) * ET_KMEANS_MAX_D;
}

namespace et {
struct ShardExtra {
    unsigned char *gathered;  // nranks records
    unsigned char *cand;      // this rank's record
    float *point;             // d floats
};
static ShardExtra shard_extra(void *workspace, int64_t N_local, int d, int K, int nranks) {
    unsigned char *p = (unsigned char *)workspace + ((et_kmeans_workspace_bytes(N_local, d, K) + 255) & ~(size_t)255);
    ShardExtra e;
    e.gathered = p;
    e.cand = p + rec_bytes(d) * (size_t)nranks;
    e.point = reinterpret_cast<float *>(p + ((rec_bytes(d) * (size_t)(nranks + 1) + 255) & ~(size_t)255));
    return e;
}
}  // namespace et

extern "C" int et_kmeans_init_farthest_sharded(const float *X, int64_t N_local, int d, int K, int64_t first_index,
                                               int64_t index_base, float *C0, float *best, void *workspace,
                                               size_t workspace_bytes, et_comm_t comm, et_stream_t stream) {
    if (!C0 || N_local < 0 || index_base < 0 || first_index < 0 || (N_local > 0 && (!X || !best))) return ET_ERR_INVALID_ARG;
    int nranks = 1;
    if (comm) {
        if (!g_rccl.ok) return ET_ERR_RCCL;
        ET_RCCL_TRY(g_rccl.CommCount((ncclComm_t)comm, &nranks));
    }
    if (!workspace || workspace_bytes < et_kmeans_sharded_workspace_bytes(N_local, d, K, nranks)) return ET_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const ShardExtra e = shard_extra(workspace, N_local, d, K, nranks);
    const size_t rb = rec_bytes(d);
    // first centroid = global point `first_index`: its owner contributes the coordinates, everybody else zeros
    ET_HIP_TRY(hipMemsetAsync(e.point, 0, sizeof(float) * d, st));
    const int64_t local = first_index - index_base;
    int rc = ET_OK;
    if (local >= 0 && local < N_local) rc = et_kmeans_gather_point(X, N_local, d, local, e.point, stream);
    if (rc) return rc;
    if (comm) ET_RCCL_TRY(g_rccl.AllReduce(e.point, e.point, (size_t)d, ncclFloat, ncclSum, (ncclComm_t)comm, st));
    rc = et_kmeans_init_set(C0, d, K, 0, e.point, stream);
    for (int i = 1; i < K && !rc; ++i) {
        rc = et_kmeans_init_step(X, N_local, d, K, i, C0, best, index_base, e.cand, workspace, workspace_bytes, stream);
        if (rc) break;
        const void *cands = e.cand;
        if (comm) {
            ET_RCCL_TRY(g_rccl.AllGather(e.cand, e.gathered, rb, ncclUint8, (ncclComm_t)comm, st));  // 32 B per rank
            cands = e.gathered;
        }
        // smallest 64-bit key wins (value first, then global index): identical on every rank
        rc = et_kmeans_init_select(cands, comm ? nranks : 1, (int)rb, d, K, i, C0, stream);
    }
    return rc;
}

extern "C" int et_internal_kmeans_chain_usable(int d, int K);
extern "C" int et_internal_kmeans_chain_run(const float *X, int64_t N, int d, int K, int max_iter, float tol, float *centroids,
                                            uint8_t *labels_u8, float *trace, et_kmeans_state *state, int64_t *partials,
                                            void *workspace, size_t workspace_bytes,
                                            int (*reduce)(void *, long long *, size_t, hipStream_t), void *ctx,
                                            et_stream_t stream);
namespace {
struct ReduceCtx {
    ncclComm_t comm;
};
int sum_over_ranks(void *ctx, long long *buf, size_t count, hipStream_t st) {  // in place, int64 SUM
    const ncclResult_t r = et::g_rccl.AllReduce(buf, buf, count, ncclInt64, ncclSum, static_cast<ReduceCtx *>(ctx)->comm, st);
    return r == ncclSuccess ? ET_OK : ET_ERR_RCCL;
}
}  // namespace

extern "C" int et_kmeans_fit_sharded(const float *X, int64_t N_local, int64_t N_total, int d, int K, int max_iter,
                                     float tol, float *centroids, int64_t *labels, float *trace,
                                     et_kmeans_state *state, uint8_t *labels_u8, int64_t *partials,
                                     et_kmeans_state *state_host, void *workspace, size_t workspace_bytes,
                                     et_comm_t comm, et_stream_t stream) {
    if (N_local < 0 || N_total < N_local || !centroids || !state || !partials || !state_host || max_iter < 1 ||
        (N_local > 0 && (!X || !labels_u8)))
        return ET_ERR_INVALID_ARG;
    if (comm && !g_rccl.ok) return ET_ERR_RCCL;
    hipStream_t st = (hipStream_t)stream;
    ncclComm_t c = (ncclComm_t)comm;
    int rc = et_kmeans_scan(X, N_local, d, state, stream);
    if (rc) return rc;
    if (comm) {  // global scale, non-finite flag and smallest non-zero |x|: identical fixed-point layout on every rank
        ET_RCCL_TRY(g_rccl.GroupStart());
        ET_RCCL_TRY(g_rccl.AllReduce(&state->max_abs_x, &state->max_abs_x, 1, ncclDouble, ncclMax, c, st));
        ET_RCCL_TRY(g_rccl.AllReduce(&state->bad_input, &state->bad_input, 1, ncclInt64, ncclMax, c, st));
        ET_RCCL_TRY(g_rccl.AllReduce(&state->min_nz_x_bits, &state->min_nz_x_bits, 1, ncclInt64, ncclMin, c, st));
        ET_RCCL_TRY(g_rccl.GroupEnd());
    }
    rc = et_kmeans_begin(state, N_total, centroids, d, K, stream);
    if (rc) return rc;
    // The chained Lloyd loop (one launch per iteration, csrc/et_kmeans.hip: km_chain_run) for the shapes it is built for
    // (d = 6, 3 <= K <= 32).  The two loop forms enqueue different collectives, so every rank must take the same one: the
    // choice depends on (d, K) alone -- a shard whose size or alignment rules out the 16-byte loads of the filter body
    // runs the exact scan inside the same chained kernel -- and needs neither a collective nor a host round trip.
    {
        const bool usable = N_total <= 0xffffffffll && et_internal_kmeans_chain_usable(d, K) != 0;
        if (usable) {
            ReduceCtx ctx{c};
            rc = et_internal_kmeans_chain_run(X, N_local, d, K, max_iter, tol, centroids, labels_u8, trace, state, partials,
                                              workspace, workspace_bytes, comm ? &sum_over_ranks : nullptr, &ctx, stream);
            if (rc) return rc;
            if (labels) {
                rc = et_kmeans_labels_i64(labels_u8, N_local, labels, stream);
                if (rc) return rc;
            }
            ET_HIP_TRY(hipMemcpyAsync(state_host, state, sizeof(et_kmeans_state), hipMemcpyDeviceToHost, st));
            ET_HIP_TRY(hipStreamSynchronize(st));
            return state_host->bad_input ? ET_ERR_BAD_DATA : ET_OK;
        }
    }
    StateRing *ring = StateRing::get(&rc);
    if (!ring) return rc;
    const size_t plen = et_kmeans_partials_len(d, K);
    constexpr int kEvery = 4;
    bool done = false;
    for (int it = 0; it < max_iter && !done; ++it) {
        rc = et_kmeans_assign_accumulate(X, N_local, d, K, state, centroids, nullptr, labels_u8, partials, workspace,
                                         workspace_bytes, stream);
        if (rc) return rc;
        if (comm) ET_RCCL_TRY(g_rccl.AllReduce(partials, partials, plen, ncclInt64, ncclSum, c, st));  // 1.1 KB, in place
        rc = et_kmeans_update(state, partials, d, K, tol, centroids, trace, stream);
        if (rc) return rc;
        // The host looks at the convergence flag one post LATE and by a blocking wait on that specific copy (posted
        // kEvery iterations ago, long since arrived): which copy a rank sees must not depend on timing, or the ranks
        // would stop enqueueing collectives at different iterations.  The flag itself is computed from identical
        // integers on every rank; launches after convergence are no-ops.
        if ((it + 1) % kEvery == 0) {
            rc = ring->post(state, st, &done);
            if (!rc && ring->pending() > 1) rc = ring->wait_oldest(&done);
            if (rc) return rc;
        }
    }
    if (labels) {
        rc = et_kmeans_labels_i64(labels_u8, N_local, labels, stream);
        if (rc) return rc;
    }
    ET_HIP_TRY(hipMemcpyAsync(state_host, state, sizeof(et_kmeans_state), hipMemcpyDeviceToHost, st));
    ET_HIP_TRY(hipStreamSynchronize(st));
    return state_host->bad_input ? ET_ERR_BAD_DATA : ET_OK;
}

extern "C" int et_internal_kmeans_reforder_sharded_run(const float *X, const int64_t *n_locals, int nranks, int rank, int K,
                                                       int max_iter, float tol, float *centroids, int64_t *labels, float *trace,
                                                       et_kmeans_state *state_host, void *workspace, size_t workspace_bytes,
                                                       int (*gather)(void *, const void *, void *, size_t, hipStream_t),
                                                       int (*agree)(void *, et_kmeans_state *, hipStream_t), void *ctx,
                                                       et_stream_t stream);
namespace {
int gather_over_ranks(void *ctx, const void *send, void *recv, size_t bytes, hipStream_t st) {
    const ncclResult_t r = et::g_rccl.AllGather(send, recv, bytes, ncclUint8, static_cast<ReduceCtx *>(ctx)->comm, st);
    return r == ncclSuccess ? ET_OK : ET_ERR_RCCL;
}
int agree_over_ranks(void *ctx, et_kmeans_state *state, hipStream_t st) {  // the scan's scale and non-finite flag: MAX
    ncclComm_t c = static_cast<ReduceCtx *>(ctx)->comm;
    if (et::g_rccl.GroupStart() != ncclSuccess) return ET_ERR_RCCL;
    const ncclResult_t r1 = et::g_rccl.AllReduce(&state->max_abs_x, &state->max_abs_x, 1, ncclDouble, ncclMax, c, st);
    const ncclResult_t r2 = et::g_rccl.AllReduce(&state->bad_input, &state->bad_input, 1, ncclInt64, ncclMax, c, st);
    if (et::g_rccl.GroupEnd() != ncclSuccess || r1 != ncclSuccess || r2 != ncclSuccess) return ET_ERR_RCCL;
    return ET_OK;
}
}  // namespace

extern "C" int et_kmeans_fit_reforder_sharded(const float *X, const int64_t *n_locals, int nranks, int rank, int d, int K,
                                              int max_iter, float tol, float *centroids, int64_t *labels, float *trace,
                                              et_kmeans_state *state_host, void *workspace, size_t workspace_bytes,
                                              et_comm_t comm, et_stream_t stream) {
    if (d != 6) return ET_ERR_UNSUPPORTED;
    if (!comm) {
        if (nranks != 1 || rank != 0) return ET_ERR_INVALID_ARG;
        return et_internal_kmeans_reforder_sharded_run(X, n_locals, 1, 0, K, max_iter, tol, centroids, labels, trace, state_host,
                                                       workspace, workspace_bytes, nullptr, nullptr, nullptr, stream);
    }
    if (!g_rccl.ok) return ET_ERR_RCCL;
    int cn = 0, cr = 0;
    ET_RCCL_TRY(g_rccl.CommCount((ncclComm_t)comm, &cn));
    ET_RCCL_TRY(g_rccl.CommUserRank((ncclComm_t)comm, &cr));
    if (cn != nranks || cr != rank) return ET_ERR_INVALID_ARG;
    ReduceCtx ctx{(ncclComm_t)comm};
    return et_internal_kmeans_reforder_sharded_run(X, n_locals, nranks, rank, K, max_iter, tol, centroids, labels, trace,
                                                   state_host, workspace, workspace_bytes, &gather_over_ranks, &agree_over_ranks,
                                                   &ctx, stream);
}
