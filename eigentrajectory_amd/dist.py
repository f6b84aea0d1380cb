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

import torch
import torch.distributed as dist

from . import ops


def _world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _all_reduce(t, op, group=None):
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=op, group=group)
    return t


# ------------------------------------------------------------------------------------------ fit
def fit_descriptor_sharded(obs, pred, k, mode, static_dist=0.0, which=1, group=None, gram_fn=None, eigh_fn=None,
                           want_count=True):
    """U_obs (2T_obs,k), U_pred (2T_pred,k), sigma_obs, sigma_pred, count for descriptor ``which`` over ALL ranks' rows.

    ``want_count=False`` skips the host read-back of the row count (the only synchronisation in here) and returns
    it as a 0-d device tensor instead."""
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

    def __init__(self, X_local, n_clusters, group=None, shard_factory=None, check_every=4):
        self.group = group
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

    # farthest-first initialisation (kmeans.py:78-112), first centroid = global point `first_index`
    def init_farthest(self, first_index):
        K, d = self.K, self.d
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

    def fit(self, centroids, max_iter=100, tol=1e-4, trace=None):
        """kmeans.py:228-240 over all ranks.  ``centroids`` (d,K) is updated in place (identical on every rank).

        Returns dict(centroids, labels (local, int64), n_iter, error, inertia, done).
        """
        sh = self.shard
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
