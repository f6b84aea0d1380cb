"""Data-sharded fit and k-means: one process per GPU, RCCL over xGMI via torch.distributed.

The reference has no distributed code at all (its "multi-GPU" is one
independent process per scene, script/train.sh:51-59).  This module adds the only
exchanges the descriptor path needs when the N trajectories are split over ranks
(SURVEY.md §8(e)); rows are independent everywhere else, so projection and
reconstruction never communicate.

* fit:      each rank accumulates its fp64 Gram matrices, ONE all-reduce(SUM) of
            2T_obs^2 + 2T_pred^2 + 1 doubles (6.7 KB), then every rank runs the same
            deterministic Jacobi solve -> identical U everywhere, no broadcast.
* k-means:  all-reduce(MAX) of max|x| once; per farthest-first step an all-gather of
            one 8+4d byte candidate per rank; per Lloyd iteration ONE all-reduce(SUM)
            of d*K+K+2 int64 (1.1 KB).  The sums are exact integers, so the result is
            bit-identical for every number of ranks and every partition.
All messages are a few KB: latency-bound, the 7 x 153 GB/s xGMI links are irrelevant here.

The compute steps are injected (``gram_fn`` / ``shard_factory``) so that the
orchestration can be exercised on CPU with the gloo backend in tests; the defaults
are the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib as L
from . import ops


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _all_reduce(t, op, group=None):
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=op, group=group)
    return t


# ------------------------------------------------------------------------------------ communicator
class Communicator:
    """An RCCL communicator (ncclComm_t) owned by libetamd.so, for the native sharded entry points
    (``et_fit_gram_sharded`` / ``et_kmeans_init_farthest_sharded`` / ``et_kmeans_fit_sharded``): their collectives are
    enqueued by the library on the caller's HIP stream, with no Python between two Lloyd iterations.

    Created collectively: rank 0 draws an ``ncclUniqueId`` and the 128 bytes travel over the already initialised
    ``torch.distributed`` group (any backend -- this is the bootstrap side channel, not the data path).  Without
    ``torch.distributed`` it is a single-rank communicator (still a real RCCL communicator: the collectives run)."""

    def __init__(self, device=None, group=None):
        self.dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.world, self.rank = _world(group)
        lib = L.lib()
        # bind the RCCL instance PyTorch ships (the one torch.distributed's "nccl" backend uses), not a second copy
        torch_rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        L.check(lib.et_comm_load(torch_rccl.encode() if os.path.exists(torch_rccl) else None), "et_comm_load")
        uid = (C.c_ubyte * 128)()
        if self.rank == 0:
            L.check(lib.et_comm_unique_id(uid), "et_comm_unique_id")
        if self.world > 1:
            box = [bytes(uid)]
            dist.broadcast_object_list(box, src=0, group=group)
            uid = (C.c_ubyte * 128).from_buffer_copy(box[0])
        self.handle = C.c_void_p()
        with torch.cuda.device(self.dev):
            L.check(lib.et_comm_init_rank(uid, self.world, self.rank, C.byref(self.handle)), "et_comm_init_rank")

    def info(self):
        n, r = C.c_int(), C.c_int()
        L.check(L.lib().et_comm_info(self.handle, C.byref(n), C.byref(r)), "et_comm_info")
        return n.value, r.value

    def close(self):
        if self.handle:
            L.check(L.lib().et_comm_destroy(self.handle), "et_comm_destroy")
            self.handle = C.c_void_p()


def fit_gram_native(obs, pred, mode, static_dist=0.0, which=1, comm=None):
    """``ops.fit_gram`` over all ranks' rows through ``et_fit_gram_sharded``: local pass + one grouped RCCL all-reduce
    on the current stream.  -> (G_obs, G_pred fp64, count int64 device tensor), identical on every rank."""
    dev = L.require_device(obs)
    obs, pred = L.on_device(obs, dev), L.on_device(pred, dev)
    n, t_obs, _ = obs.shape
    t_pred = pred.shape[1]
    g_obs = torch.empty((2 * t_obs, 2 * t_obs), device=dev, dtype=torch.float64)
    g_pred = torch.empty((2 * t_pred, 2 * t_pred), device=dev, dtype=torch.float64)
    count = torch.zeros((1,), device=dev, dtype=torch.int64)
    ws = torch.empty((max(L.lib().et_fit_gram_workspace_bytes(L.i64(n), t_obs, t_pred), 8),), device=dev, dtype=torch.uint8)
    L.check(L.lib().et_fit_gram_sharded(L.ptr(obs), L.ptr(pred), L.i64(n), t_obs, t_pred, int(mode), L.f32(static_dist),
                                        int(which), L.ptr(g_obs), L.ptr(g_pred), L.ptr(count), L.ptr(ws),
                                        C.c_size_t(ws.numel()), comm.handle if comm is not None else None, L.stream(dev)),
            "et_fit_gram_sharded")
    return g_obs, g_pred, count


# ------------------------------------------------------------------------------------------ fit
def fit_descriptor_sharded(obs, pred, k, mode, static_dist=0.0, which=1, group=None, gram_fn=None, eigh_fn=None,
                           want_count=True, comm=None):
    """U_obs (2T_obs,k), U_pred (2T_pred,k), sigma_obs, sigma_pred, count for descriptor ``which`` over ALL ranks' rows.

    ``want_count=False`` skips the host read-back of the row count (the only synchronisation in here) and returns
    it as a 0-d device tensor instead."""
    if comm is not None:  # native: the library enqueues the grouped all-reduce itself
        g_obs, g_pred, cnt = fit_gram_native(obs, pred, mode, static_dist, which, comm)
        count = int(cnt.item()) if want_count else cnt[0]
    else:
        gram_fn = gram_fn or ops.fit_gram
        g_obs, g_pred, cnt = gram_fn(obs, pred, mode, static_dist, which)
        packed = torch.cat([g_obs.reshape(-1), g_pred.reshape(-1), cnt.reshape(-1).to(g_obs.dtype)])
        _all_reduce(packed, dist.ReduceOp.SUM, group)  # one 6.7 KB message
        no, npd = g_obs.numel(), g_pred.numel()
        g_obs = packed[:no].reshape(g_obs.shape).contiguous()
        g_pred = packed[no:no + npd].reshape(g_pred.shape).contiguous()
        count = int(round(float(packed[-1].item()))) if want_count else packed[-1]
    if eigh_fn is None:
        (U_obs, s_obs), (U_pred, s_pred) = ops.eigh_topk_batch([g_obs, g_pred], k)  # both matrices in one launch
    else:
        U_obs, s_obs = eigh_fn(g_obs, k)
        U_pred, s_pred = eigh_fn(g_pred, k)
    return U_obs, U_pred, s_obs, s_pred, count


# -------------------------------------------------------------------------------------- k-means
class ShardedKMeans:
    """BatchKMeans (one batch element) over points split across ranks.

    ``shard`` exposes the step interface of :class:`eigentrajectory_amd.ops.KMeansShard`
    (scan / begin / init_step / gather_point / assign / update / labels / read_state and the
    ``state``, ``state_f64`` tensors).  Every rank must call the same methods in the same order.
    """

    def __init__(self, X_local, n_clusters, group=None, shard_factory=None, check_every=4, comm=None):
        self.group = group
        self.comm = comm  # a Communicator: init_farthest / fit then run as ONE native call each (no Python in the loop)
        self.world, self.rank = _world(group)
        self.K = int(n_clusters)
        self.shard = (shard_factory or ops.KMeansShard)(X_local, self.K)
        self.d, self.n_local = self.shard.d, self.shard.n
        self.dev = self.shard.state.device
        counts = torch.zeros((self.world,), dtype=torch.int64, device=self.dev)
        counts[self.rank] = self.n_local
        _all_reduce(counts, dist.ReduceOp.SUM, group)
        counts = counts.cpu().tolist()
        self.n_total = int(sum(counts))
        self.index_base = int(sum(counts[:self.rank]))
        self.counts = counts
        self.check_every = int(check_every)
        if self.n_total >= 2 ** 32:
            raise ValueError("farthest-first keys carry 32-bit global indices: N_total must be < 2^32")

    def _native_workspace(self):
        if not hasattr(self, "_ws_native"):
            nbytes = L.lib().et_kmeans_sharded_workspace_bytes(L.i64(self.n_local), self.d, self.K, self.world)
            self._ws_native = torch.empty((nbytes,), device=self.dev, dtype=torch.uint8)
        return self._ws_native

    # farthest-first initialisation (kmeans.py:78-112), first centroid = global point `first_index`
    def init_farthest(self, first_index):
        K, d = self.K, self.d
        if self.comm is not None:
            sh, ws = self.shard, self._native_workspace()
            C0 = torch.zeros((d, K), dtype=torch.float32, device=self.dev)
            L.check(L.lib().et_kmeans_init_farthest_sharded(
                L.ptr(sh.X), L.i64(self.n_local), d, K, L.i64(first_index), L.i64(self.index_base), L.ptr(C0), L.ptr(sh.best),
                L.ptr(ws), C.c_size_t(ws.numel()), self.comm.handle, L.stream(self.dev)), "et_kmeans_init_farthest_sharded")
            return C0
        C0 = torch.zeros((d, K), dtype=torch.float32, device=self.dev)
        first = torch.zeros((d,), dtype=torch.float32, device=self.dev)
        local = int(first_index) - self.index_base
        if 0 <= local < self.n_local:
            first = self.shard.gather_point(local).to(torch.float32)
        _all_reduce(first, dist.ReduceOp.SUM, self.group)  # only the owner contributes non-zeros
        C0[:, 0] = first
        rec_bytes = (8 + 4 * d + 7) // 8 * 8
        gathered = torch.zeros((self.world * rec_bytes,), dtype=torch.uint8, device=self.dev)
        for i in range(1, K):
            cand = self.shard.init_step(i, C0, self.index_base)[:rec_bytes]
            if dist.is_available() and dist.is_initialized():
                dist.all_gather_into_tensor(gathered, cand, group=self.group)  # 32 B per rank
            else:
                gathered.copy_(cand)
            # smallest 64-bit key wins (value first, then global index): identical on every rank, one tiny kernel
            self.shard.init_select(gathered, self.world, rec_bytes, i, C0)
        return C0

    def fit(self, centroids, max_iter=100, tol=1e-4, trace=None, sums="exact"):
        """kmeans.py:228-240 over all ranks.  ``centroids`` (d,K) is updated in place (identical on every rank).

        sums="reference-order": the reference's own fp32 summation orders (ops.kmeans_fit_reference_order on the whole
        array, bit for bit); needs the shard sizes of ops.reference_order_shard_sizes and, beyond one rank, a Communicator.

        Returns dict(centroids, labels (local, int64), n_iter, error, inertia, done).
        """
        sh = self.shard
        if sums == "reference-order":
            if self.world > 1 and self.comm is None:
                raise NotImplementedError("sums='reference-order' over several ranks runs as one native call: pass comm=Communicator(...)")
            res = ops.kmeans_fit_reference_order_sharded(sh.X, centroids, self.counts, self.rank, self.comm, max_iter, tol,
                                                         trace=trace is not None)
            if trace is not None:  # (the caller's (max_iter, 2) tensor, like the exact path: rows [0, n_iter) are written)
                trace[:res["n_iter"]].copy_(res["trace"])
            centroids.copy_(res["centroids"])
            res["centroids"] = centroids
            return res
        if sums != "exact":
            raise ValueError(f"sums must be 'exact' or 'reference-order', got {sums!r}")
        if self.comm is not None:
            ws = self._native_workspace()
            st = L.KMeansState()
            labels = torch.empty((self.n_local,), device=self.dev, dtype=torch.int64)
            L.check(L.lib().et_kmeans_fit_sharded(
                L.ptr(sh.X), L.i64(self.n_local), L.i64(self.n_total), self.d, self.K, int(max_iter), L.f32(tol),
                L.ptr(centroids), L.ptr(labels), L.ptr(trace), L.ptr(sh.state), L.ptr(sh.labels_u8), L.ptr(sh.partials),
                C.byref(st), L.ptr(ws), C.c_size_t(ws.numel()), self.comm.handle, L.stream(self.dev)),
                "et_kmeans_fit_sharded")
            return dict(centroids=centroids, labels=labels, n_iter=int(st.iter), error=float(st.error),
                        inertia=float(st.inertia), done=bool(st.done))
        sh.scan()
        # global max|x| and non-finite flag
        _all_reduce(sh.state_f64[0:1], dist.ReduceOp.MAX, self.group)
        _all_reduce(sh.state[7:8], dist.ReduceOp.MAX, self.group)
        _all_reduce(sh.state[11:12], dist.ReduceOp.MIN, self.group)  # smallest non-zero |x| (fp32 bits)
        sh.begin(self.n_total, centroids)
        # Convergence is decided on the device from identical bits on every rank.  The host looks at the state block
        # `check_every` iterations LATE (an asynchronous copy posted then, long since arrived): no pipeline drain,
        # and every rank takes the same decision at the same iteration.  Launches after convergence are no-ops.
        pending = []
        st = None
        for it in range(max_iter):
            part = sh.assign(centroids)  # the shard keeps its running totals; `part` is a copy we may overwrite
            _all_reduce(part, dist.ReduceOp.SUM, self.group)  # 1.1 KB of int64, in place
            sh.update(part, centroids, tol, trace)
            if (it + 1) % self.check_every == 0:
                pending.append(sh.post_state())
                if len(pending) > 1:
                    st = sh.wait_state(pending.pop(0))
                    if st.done:
                        break
        st = sh.read_state()
        if st.bad_input:
            raise ValueError("k-means input contains NaN/Inf")
        return dict(centroids=centroids, labels=sh.labels(), n_iter=int(st.iter), error=float(st.error),
                    inertia=float(st.inertia), done=bool(st.done))
