"""world_size-2 gloo runs (CPU) of the sharded fit / k-means orchestration in eigentrajectory_amd.dist.

The compute steps are injected: here they are backed by the CPU oracle (tests may use it; the
product never does), so what is exercised is the exchange logic -- which tensors are reduced, in
what order, and that every rank ends with the bit-identical result a single-shard run gives.
"""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from . import _golden as G


class OracleShard:
    """ops.KMeansShard look-alike on CPU tensors, backed by oracle/et_oracle.c."""

    def __init__(self, X, K):
        from oracle import et_oracle as eo
        self.eo = eo
        self.X = np.ascontiguousarray(X.numpy() if isinstance(X, torch.Tensor) else X, dtype=np.float32)
        self.d, self.n = self.X.shape
        self.K = K
        self.state = torch.zeros((12,), dtype=torch.int64)
        self.partials = torch.zeros((self.d * K + K + 2,), dtype=torch.int64)
        self._labels = np.zeros((self.n,), np.int64)
        self.best = np.zeros((self.n,), np.float32)
        self.cand = torch.zeros((8 + 4 * 32,), dtype=torch.uint8)

    @property
    def state_f64(self):
        return self.state.view(torch.float64)

    def scan(self):
        self.state.zero_()
        self.state_f64[0] = float(np.abs(self.X).max()) if self.n else 0.0
        self.state[7] = int(not np.isfinite(self.X).all())
        nz = np.abs(self.X[self.X != 0])
        self.state[11] = int(nz.min().view(np.uint32)) if nz.size else 0x7f800000

    def begin(self, n_total, centroids):
        mx = float(self.state_f64[0])
        self.state[2] = n_total
        self.state[3] = self.eo.kmeans_frac_bits(mx, n_total)
        mc = float(np.nanmax(np.abs(centroids.numpy()))) if np.isfinite(centroids.numpy()).any() else 0.0
        self.state_f64[1] = mc
        self.state[4] = self.eo.kmeans_sim_frac_bits(mx, mc, self.d, n_total)
        self.state[5] = 0
        self.state[6] = int(self.state[7])

    def gather_point(self, i):
        return torch.from_numpy(self.X[:, i].copy())

    def init_step(self, i, C0, index_base):
        c = C0.numpy()[:, i - 1]
        key = np.uint64(0xFFFFFFFFFFFFFFFF)
        pt = np.full((self.d,), np.nan, np.float32)
        if self.n:
            y = self.eo.euc_sim(self.X, c[:, None].copy())[:, 0]
            self.best = y if i == 1 else np.where((y > self.best) | (np.isnan(y) & ~np.isnan(self.best)), y, self.best)
            b = self.best
            bits = b.view(np.uint32).astype(np.uint64)
            order = np.where(bits & np.uint64(0x80000000), (~bits) & np.uint64(0xFFFFFFFF), bits | np.uint64(0x80000000))
            order = np.where(np.isnan(b), np.uint64(0), order)
            order = np.where(b == 0, np.uint64(0x80000000), order)
            keys = (order << np.uint64(32)) | (np.arange(self.n, dtype=np.uint64) + np.uint64(index_base))
            w = int(np.argmin(keys))
            key, pt = keys[w], self.X[:, w]
        buf = np.zeros((8 + 4 * 32,), np.uint8)
        buf[:8] = np.frombuffer(np.uint64(key).tobytes(), np.uint8)
        buf[8:8 + 4 * self.d] = np.frombuffer(np.ascontiguousarray(pt, np.float32).tobytes(), np.uint8)
        self.cand = torch.from_numpy(buf)
        return self.cand

    def init_select(self, cands, n_cands, stride, col, C0):
        recs = cands.numpy().reshape(n_cands, stride)
        keys = recs[:, :8].copy().view(np.uint64).reshape(-1)
        w = int(np.argmin(keys))
        C0[:, col] = torch.from_numpy(recs[w, 8:8 + 4 * self.d].copy().view(np.float32))

    def post_state(self):
        return self.read_state()

    def wait_state(self, handle):
        return handle

    def assign(self, centroids, given_labels=None):
        if int(self.state[6]):
            return self.partials
        if self.n:
            lb, s, c, ss, nn = self.eo.kmeans_assign_accumulate(self.X, centroids.numpy(), int(self.state[3]),
                                                                int(self.state[4]))
            self._labels = lb
            flat = np.concatenate([s.reshape(-1), c, [ss, nn]]).astype(np.int64)
        else:
            flat = np.zeros((self.d * self.K + self.K + 2,), np.int64)
        self.partials = torch.from_numpy(flat)
        return self.partials

    def update(self, partials, centroids, tol, trace=None):
        if int(self.state[6]):
            return
        p = partials.numpy()
        d, K = self.d, self.K
        n_total = int(self.state[2])
        c_new, err, ine, done = self.eo.kmeans_update(p[:d * K].reshape(d, K), p[d * K:d * K + K], int(p[-2]), int(p[-1]),
                                                      n_total, int(self.state[3]), int(self.state[4]), tol,
                                                      centroids.numpy())
        centroids.copy_(torch.from_numpy(c_new))
        mc = float(np.nanmax(np.abs(c_new))) if np.isfinite(c_new).any() else 0.0
        self.state_f64[1] = mc
        self.state[4] = self.eo.kmeans_sim_frac_bits(float(self.state_f64[0]), mc, d, n_total)
        self.state_f64[8] = err
        self.state_f64[9] = ine
        self.state[5] += 1
        self.state[6] = int(done)

    def labels(self):
        return torch.from_numpy(self._labels.copy())

    def read_state(self):
        from eigentrajectory_amd import _lib
        return _lib.KMeansState.from_buffer_copy(self.state.numpy().tobytes())


def oracle_gram(obs, pred, mode, static_dist, which):
    from oracle import et_oracle as eo
    g_obs, g_pred, cnt = eo.fit_gram(obs.numpy(), pred.numpy(), mode, static_dist, which)
    return torch.from_numpy(g_obs), torch.from_numpy(g_pred), torch.tensor([cnt], dtype=torch.int64)


def oracle_eigh(G_, k):
    from oracle import et_oracle as eo
    U, s = eo.eigh_topk(G_.numpy(), k)
    return torch.from_numpy(U), torch.from_numpy(s)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, cuts, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from eigentrajectory_amd.dist import ShardedKMeans, fit_descriptor_sharded
        from eigentrajectory_amd.synth import gaussian_points_np, synthetic_trajectories_np
        obs, pred = synthetic_trajectories_np(4000, seed=1)
        lo, hi = cuts[rank], cuts[rank + 1]
        U_obs, U_pred, s_obs, s_pred, count = fit_descriptor_sharded(
            torch.from_numpy(obs[lo:hi]), torch.from_numpy(pred[lo:hi]), 6, 2, 0.3, 1, gram_fn=oracle_gram,
            eigh_fn=oracle_eigh)
        x = gaussian_points_np(6, 4000, seed=2, n_blobs=6)
        km = ShardedKMeans(torch.from_numpy(np.ascontiguousarray(x[:, lo:hi])), 20, shard_factory=OracleShard,
                           check_every=3)
        c0 = km.init_farthest(first_index=1234)
        res = km.fit(c0.clone(), max_iter=40, tol=1e-4)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), U_obs=U_obs.numpy(), U_pred=U_pred.numpy(), count=count,
                 c0=c0.numpy(), centroids=res["centroids"].numpy(), labels=res["labels"].numpy(), n_iter=res["n_iter"],
                 inertia=res["inertia"], n_total=km.n_total, index_base=km.index_base)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("cuts", [(0, 2000, 4000), (0, 37, 4000)])
def test_sharded_fit_and_kmeans_two_ranks_gloo(tmp_path, oracle, cuts):
    from eigentrajectory_amd.synth import gaussian_points_np, synthetic_trajectories_np
    mp.spawn(_worker, args=(2, _free_port(), cuts, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    # every rank holds the same fit and the same centroids
    for key in ("U_obs", "U_pred", "count", "c0", "centroids", "n_iter", "inertia", "n_total"):
        assert np.array_equal(r0[key], r1[key], equal_nan=True), key
    assert int(r0["n_total"]) == 4000 and int(r1["index_base"]) == cuts[1]
    # ... and it is what one process computes on the whole data
    obs, pred = synthetic_trajectories_np(4000, seed=1)
    g_obs, g_pred, cnt = oracle.fit_gram(obs, pred, 2, 0.3, 1)
    assert int(r0["count"]) == cnt
    U_ref, _ = oracle.eigh_topk(g_pred, 6)
    np.testing.assert_allclose(r0["U_pred"], U_ref, atol=1e-6)  # fp64 partial sums in another order
    x = gaussian_points_np(6, 4000, seed=2, n_blobs=6)
    c0, _ = oracle.kmeans_init_farthest(x, 20, 1234)
    assert np.array_equal(r0["c0"], c0)
    ref = oracle.kmeans_fit(x, c0, 40, 1e-4)
    assert int(r0["n_iter"]) == ref["n_iter"]
    assert np.array_equal(r0["centroids"], ref["centroids"])          # exact integer sums: bit-identical
    assert np.array_equal(np.concatenate([r0["labels"], r1["labels"]]), ref["labels"])


def _reforder_exchange_worker(rank, world, port, sizes, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ._reforder_shard_np import level_step, replay, shard_record
        n = sum(sizes)
        rng = np.random.RandomState(7)
        d, K = 6, 5
        x = (rng.standard_normal((d, n)) * np.array([[1], [30], [1e-2], [1], [5], [1]])).astype(np.float32)
        lab = rng.randint(K, size=n)
        lab[rng.rand(n) < 0.3] = 2  # uneven clusters, one of them big
        lo = sum(sizes[:rank])
        L = level_step(n)
        rec = shard_record(x[:, lo:lo + sizes[rank]], lab[lo:lo + sizes[rank]], K, L)
        # the record as the kernels lay it out: max_rows rows | T1 | T0 | leftovers (3 points) | full_rows
        block = 4 * L ** 3
        max_rows = max(1, max(-(-(s // (4 * L * L)) // L) for s in sizes))
        rows = np.zeros((max_rows, d, K, 4), np.float32)
        rows[:len(rec["rows"])] = rec["rows"]
        left = np.zeros((d, K, 3), np.float32)
        left[:, :, :rec["left"].shape[2]] = rec["left"]
        flat = np.concatenate([rows.ravel(), rec["T1"].ravel(), rec["T0"].ravel(), left.ravel(),
                               np.asarray([len(rec["rows"]), rec["full_rows"], rec["left"].shape[2]], np.float32)])
        mine = torch.from_numpy(flat)
        table = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(table, mine)  # ONE all-gather per iteration: what ncclAllGather does in et_kmeans_fit_reforder_sharded
        records = []
        for t in table:
            a = t.numpy()
            o = 0
            r_ = a[o:o + rows.size].reshape(rows.shape); o += rows.size
            t1 = a[o:o + d * K * 4].reshape(d, K, 4); o += d * K * 4
            t0 = a[o:o + d * K * 4].reshape(d, K, 4); o += d * K * 4
            lf = a[o:o + d * K * 3].reshape(d, K, 3); o += d * K * 3
            nrows, nfull, nleft = (int(v) for v in a[o:o + 3])
            records.append(dict(rows=r_[:nrows], full_rows=nfull, T1=t1, T0=t0, left=lf[:, :, :nleft]))
        tail_rank = max(i for i, s in enumerate(sizes) if s > 0)
        assert all(s % block == 0 for s in sizes[:tail_rank])
        sums = replay(records, tail_rank)
        np.save(os.path.join(out_dir, f"sums{rank}.npy"), sums)
        if rank == 0:
            np.savez(os.path.join(out_dir, "data.npz"), x=x, lab=lab, K=K)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sizes", [(16384, 23616), (32768, 7235), (40000, 0)])
def test_reference_order_shard_exchange_two_ranks_gloo(tmp_path, oracle, sizes):
    """The exchange of the sharded reference-order k-means (et_kmeans_fit_reforder_sharded) with two gloo ranks on CPU: every
    rank restates its levels 0 .. 2 in numpy (tests/_reforder_shard_np.py: the kernels' additions in the kernels' order),
    ONE all-gather of the records, the same level-3 replay on both ranks -> the cluster sums of the oracle's literal ATen-order
    sum over the WHOLE array, bit for bit, on every rank.  Shards: the end of the array inside the second shard / a shard
    that ends on N mod 4 != 0 / an empty trailing rank."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_reforder_exchange_worker, args=(2, port, sizes, str(tmp_path)), nprocs=2, join=True)
    z = np.load(tmp_path / "data.npz")
    ref = oracle.kmeans_reforder_sums(z["x"], z["lab"], int(z["K"]))
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"sums{r}.npy"), ref), r
