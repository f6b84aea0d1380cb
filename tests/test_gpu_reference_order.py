"""GPU parity tests -- BatchKMeans(sums="reference-order"): ATen's own summation orders, whole reference runs bit for bit.
HIP path (through the C ABI, via eigentrajectory_amd.ops) vs the CPU oracle and the golden vectors; needs a real MI355X:
run with ``pytest -m gpu``."""
import os

import numpy as np
import pytest
import torch

from . import _golden as G
from ._gpu_common import *  # noqa: F401,F403 -- fixtures (dev, ops) and helpers

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,blocks", [(40000, (1, 1)), (40000, (2, 0)), (40000, (1, 1, 1)), (100000, (2, 2, 2, 1)),
                                      (32768, (1, 1)), (5000, (1, 0)), (4300000, (16, 17)), (4300000, (32, 1, 0))])
def test_reference_order_shards_equal_the_single_gpu_fit(ops, dev, oracle, n, blocks):
    """sums="reference-order" over shards cut at level-2 blocks of ATen's cascade: every shard ends with the centroids,
    the error trace and the iteration count of the single-GPU reference-order fit on the whole array, bit for bit, and
    the shards' labels together are its labels; for the small cases that fit is checked against the oracle as well.
    Cases: the end of the array inside the last shard's first block / nothing left for the last shard / an empty trailing
    rank / N a multiple of a block / L = 32 (4.3e6 points: blocks of 131 072)."""
    import ctypes as C
    from eigentrajectory_amd import _lib as L
    from eigentrajectory_amd.synth import gaussian_points_np
    K, max_iter, tol = 20, 12, 1e-4
    x = gaussian_points_np(6, n, seed=17, n_blobs=9)
    x[:, ::61] *= 25.0
    block = int(L.lib().et_kmeans_reforder_shard_block(L.i64(n), 6, K))
    assert block == (16384 if n <= 4 << 19 else 131072)
    sizes, left = [], n
    for b in blocks:
        sizes.append(min(left, b * block))
        left -= sizes[-1]
    sizes[[i for i, b in enumerate(blocks) if b][-1]] += left  # the last non-empty shard takes the end of the array
    assert sum(sizes) == n
    c0 = N_(ops.kmeans_init_farthest_reference_order(T(x, dev), K, 5))
    whole = ops.kmeans_fit_reference_order(T(x, dev), T(c0, dev), max_iter, tol)
    if n <= 100000:
        ref = oracle.kmeans_fit(x, c0, max_iter, tol, sums="reference-order")
        assert np.array_equal(N_(whole["centroids"]), ref["centroids"]) and np.array_equal(N_(whole["labels"]), ref["labels"])
    shards = _reference_order_shards_native(dev, x, c0, sizes, max_iter, tol)
    for r, sh in enumerate(shards):
        assert np.array_equal(sh["centroids"], N_(whole["centroids"])), r
        assert int(sh["state"].iter) == whole["n_iter"] and bool(sh["state"].done) == whole["done"]
        assert np.float32(sh["state"].error) == np.float32(whole["error"])
        assert np.array_equal(sh["trace"][:, 0], N_(whole["trace"])[:, 0])
        np.testing.assert_allclose(sh["trace"][:, 1], N_(whole["trace"])[:, 1], rtol=1e-6)  # (the inertia: fp32 rounding of an fp64 sum)
    assert np.array_equal(np.concatenate([sh["labels"] for sh in shards]), N_(whole["labels"]))


def test_reference_order_sharded_entry_point_one_rank(ops, dev):
    """et_kmeans_fit_reforder_sharded without a communicator (one rank: the record is copied instead of gathered) and
    the shard-size helper; sizes that cut inside a block are refused."""
    from eigentrajectory_amd.synth import gaussian_points_np
    x = gaussian_points_np(6, 50000, seed=3, n_blobs=6)
    c0 = x[:, :20].copy()
    whole = ops.kmeans_fit_reference_order(T(x, dev), T(c0, dev), 10, 1e-4)
    one = ops.kmeans_fit_reference_order_sharded(T(x, dev), T(c0, dev), [50000], 0, None, 10, 1e-4)
    assert torch.equal(one["centroids"], whole["centroids"]) and torch.equal(one["labels"], whole["labels"])
    assert one["n_iter"] == whole["n_iter"] and torch.equal(one["trace"][:, 0], whole["trace"][:, 0])
    assert ops.reference_order_shard_sizes(50000, 2) == [32768, 17232]
    assert ops.reference_order_shard_sizes(50000, 8) == [16384, 16384, 16384, 848, 0, 0, 0, 0]
    with pytest.raises(NotImplementedError):
        ops.kmeans_fit_reference_order_sharded(T(x[:, :25000], dev), T(c0, dev), [25000, 25000], 0, None, 10, 1e-4)


@pytest.mark.parametrize("n,K,kind", [(1024, 20, 0), (5003, 32, 1), (30000, 20, 2), (30000, 8, 3), (70001, 20, 4), (200000, 20, 2)])
def test_reference_order_farthest_first_with_the_point_skip_vs_oracle(ops, dev, oracle, et_option, n, K, kind):
    """kmeans.py:88-112 in the reference's orders with the big-shard form forced on (reforder_init_skip_min = 0: a point whose
    nearest centroid is closer than half the way to the new one is not read): the oracle's literal picks -- every step's
    euc_sim against ALL current centroids -- bit for bit, on blobs, outliers x 1000 (new centroids far from everything: most
    points skip), far-from-origin data (the error bound E dominates: nothing skips), duplicated points (ties), NaN-free
    tiny scales; and the same picks with the skip off."""
    rng = np.random.RandomState(100 + kind)
    x = rng.standard_normal((6, n)).astype(np.float32)
    if kind == 1:
        x += np.float32(200.0)
    elif kind == 2:
        x[:, ::97] *= np.float32(1000.0)
    elif kind == 3:
        x[:, n // 2:] = x[:, :n - n // 2]
    elif kind == 4:
        x *= np.float32(1e-12)
    first = int(rng.randint(n))
    ref, _ = oracle.kmeans_init_farthest(x, K, first, reference_order=True)
    et_option("reforder_init_skip_min", 0)
    got = N_(ops.kmeans_init_farthest_reference_order(T(x, dev), K, first))
    et_option("reforder_init_skip_min", 1 << 40)
    plain = N_(ops.kmeans_init_farthest_reference_order(T(x, dev), K, first))
    assert np.array_equal(got, ref) and np.array_equal(plain, ref)


# ------------------------------------------------ BatchKMeans in the reference's summation orders (opt-in mode)
@pytest.mark.parametrize("n,filter_lp", [(1000, 9), (10000, 9), (100000, 9), (10000, 4), (100000, 4)])
def test_reference_order_kmeans_g7c(ops, dev, et_option, n, filter_lp):
    """Whole runs of the imported reference's BatchKMeans (tests/golden/g7c: 32 data sets per size, farthest-first seeding +
    <= 100 Lloyd iterations).  `sums="reference-order"`: the product ends EVERY run with the reference's initial centroids,
    labels, iteration count and final centroid bits.  Default (exact sums): same initial centroids; the whole-run
    equality rate is what it is -- 32/32, 31/32, 13/32 -- and is asserted so that it cannot drift unnoticed."""
    import hashlib
    from eigentrajectory_amd.synth import gaussian_points_np
    et_option("reforder_filter_min_lp", filter_lp)  # 4: labels certified by the matrix-core filter from the second iteration on
    z = G.load("g7c_batchkmeans_seeds.npz")
    equal_exact = 0
    for seed in z["seeds"]:
        tag = f"n{n}.s{int(seed)}"
        x = T(gaussian_points_np(6, n, seed=int(seed), n_blobs=int(z[f"{tag}.blobs"])), dev)
        first = int(z[f"{tag}.first_index"])
        c0 = ops.kmeans_init_farthest_reference_order(x, 20, first)
        assert np.array_equal(N_(c0), z[f"{tag}.c0"]), tag
        assert torch.equal(ops.kmeans_init_farthest(x, 20, first), c0)
        r = ops.kmeans_fit_reference_order(x, c0, 100, 1e-4)
        lab = N_(r["labels"]).astype(np.uint8)
        assert r["n_iter"] == int(z[f"{tag}.n_iter"]), tag
        assert hashlib.sha256(lab.tobytes()).digest() == bytes(z[f"{tag}.labels_sha256"]), tag
        assert np.array_equal(N_(r["centroids"]), z[f"{tag}.centroids"]), tag
        assert N_(r["trace"])[-1, 0] == np.float32(z[f"{tag}.final_error_inertia"][0])
        np.testing.assert_allclose(r["inertia"], z[f"{tag}.final_error_inertia"][1], rtol=1e-5)
        e = ops.kmeans_fit(x, c0, 100, 1e-4, trace=False)
        equal_exact += (e["n_iter"] == int(z[f"{tag}.n_iter"]) and
                        hashlib.sha256(N_(e["labels"]).astype(np.uint8).tobytes()).digest() == bytes(z[f"{tag}.labels_sha256"]))
    assert equal_exact == {1000: 32, 10000: 31, 100000: 13}[n]


@pytest.mark.parametrize("filter_lp", [9, 4])
def test_reference_order_kmeans_1e6_g7d(ops, dev, et_option, filter_lp):
    """Whole runs of the imported reference's BatchKMeans at N = 1e6 (tests/golden/g7d, tools/make_golden_batchkmeans_1e6.py:
    eight data sets, farthest-first seeding + <= 100 Lloyd iterations on one CPU thread, ~80 s each).  The reference-order
    fit ends EVERY run with the reference's initial centroids, iteration count, labels and final centroid bits; the default
    (exact sums) starts from the same centroids and its whole-run equality rate is what it is -- asserted so that it cannot
    drift unnoticed."""
    import hashlib
    from eigentrajectory_amd.synth import gaussian_points_np
    et_option("reforder_filter_min_lp", filter_lp)  # 4: with the matrix-core label certification (off by default)
    z = G.load("g7d_batchkmeans_1e6.npz")
    n = int(z["sizes"][0])
    equal_exact = 0
    for seed in z["seeds"]:
        tag = f"n{n}.s{int(seed)}"
        x = T(gaussian_points_np(6, n, seed=int(seed), n_blobs=int(z[f"{tag}.blobs"])), dev)
        first = int(z[f"{tag}.first_index"])
        c0 = ops.kmeans_init_farthest_reference_order(x, 20, first)
        assert np.array_equal(N_(c0), z[f"{tag}.c0"]), tag
        assert torch.equal(ops.kmeans_init_farthest(x, 20, first), c0)
        r = ops.kmeans_fit_reference_order(x, c0, 100, 1e-4)
        lab = N_(r["labels"]).astype(np.uint8)
        assert r["n_iter"] == int(z[f"{tag}.n_iter"]), tag
        assert hashlib.sha256(lab.tobytes()).digest() == bytes(z[f"{tag}.labels_sha256"]), tag
        if f"{tag}.labels" in z.files:
            assert np.array_equal(lab, z[f"{tag}.labels"])
        assert np.array_equal(np.bincount(lab, minlength=20), z[f"{tag}.counts"])
        assert np.array_equal(N_(r["centroids"]), z[f"{tag}.centroids"]), tag
        assert N_(r["trace"])[-1, 0] == np.float32(z[f"{tag}.final_error_inertia"][0])
        np.testing.assert_allclose(r["inertia"], z[f"{tag}.final_error_inertia"][1], rtol=1e-5)
        e = ops.kmeans_fit(x, c0, 100, 1e-4, trace=False)
        equal_exact += (e["n_iter"] == int(z[f"{tag}.n_iter"]) and
                        hashlib.sha256(N_(e["labels"]).astype(np.uint8).tobytes()).digest() == bytes(z[f"{tag}.labels_sha256"]))
        del x
    print(f"exact sums end with the reference's labels in {equal_exact} of {len(z['seeds'])} runs at N = 1e6")
    assert equal_exact == G7D_EXACT_EQUAL


@pytest.mark.parametrize("n,d,K", [(1, 6, 1), (5, 6, 3), (37, 6, 7), (1003, 6, 20), (4099, 6, 33), (20000, 6, 20), (777, 9, 5),
                                   (3001, 17, 40), (64, 32, 255)])
def test_reference_order_ops_vs_oracle(ops, oracle, dev, n, d, K):
    """euc_sim / predict / farthest-first / fit of the reference-order mode against the oracle's restatement (itself pinned
    against torch, tests/test_oracle_golden.py::test_reforder_arithmetic_equals_torch): every bit, any d, K, N -- the
    order of a norm depends on the column's position, so odd sizes are the point."""
    rng = np.random.RandomState(n + d + K)
    x = (rng.standard_normal((d, n)) * 2 + 0.5).astype(np.float32)
    x[:, ::13] *= 7.0
    Kc = min(K, n)
    first = int(rng.randint(n))
    c0_ref, _ = oracle.kmeans_init_farthest(x, Kc, first, reference_order=True)
    X = T(x, dev)
    c0 = ops.kmeans_init_farthest_reference_order(X, Kc, first)
    assert np.array_equal(N_(c0), c0_ref)
    assert np.array_equal(N_(ops.euc_sim_reference_order(X, c0)), oracle.euc_sim(x, c0_ref, reference_order=True))
    lab_ref, ms_ref = oracle.kmeans_assign(x, c0_ref, reference_order=True)
    lab, ms = ops.kmeans_predict_reference_order(X, c0)
    assert np.array_equal(N_(lab), lab_ref) and np.array_equal(N_(ms), ms_ref)
    ref = oracle.kmeans_fit(x, c0_ref, 30, 1e-4, sums="reference-order")
    res = ops.kmeans_fit_reference_order(X, c0, 30, 1e-4)
    assert res["n_iter"] == ref["n_iter"]
    assert np.array_equal(N_(res["labels"]), ref["labels"])
    assert np.array_equal(N_(res["centroids"]), ref["centroids"], equal_nan=True)
    assert np.array_equal(N_(res["trace"])[:, 0], ref["trace"][:, 0], equal_nan=True)
    np.testing.assert_allclose(N_(res["trace"])[:, 1], ref["trace"][:, 1], rtol=1e-5, equal_nan=True)


def test_reference_order_batchkmeans_module_gauss10000(dev):
    """The G7 case the exact-sum fit does not reproduce (another local optimum): BatchKMeans(sums="reference-order") ends
    with the reference's labels and centroid bits; the class draws its first centroid where the reference does."""
    from eigentrajectory_amd import BatchKMeans
    from eigentrajectory_amd.synth import gaussian_points_np
    z = G.load("g7_batchkmeans.npz")
    x = T(gaussian_points_np(6, 10000, seed=11, n_blobs=0), dev)[None].contiguous()
    km = BatchKMeans(n_clusters=20, max_iter=100, tol=1e-4, sums="reference-order")
    km.rng = np.random.RandomState(0)
    labels = km.fit(x)
    assert np.array_equal(N_(labels[0]), z["gauss10000.labels"].astype(np.int64))
    assert np.array_equal(N_(km.centroids[0]), z["gauss10000.centroids"])
    assert km.n_iter_ == [len(z["gauss10000.trace"])]
    ql = km.predict(T(gaussian_points_np(6, 512, seed=12, n_blobs=0), dev)[None].contiguous())
    assert np.array_equal(N_(ql[0]), z["gauss10000.query_labels"].astype(np.int64))
    with pytest.raises(NotImplementedError):  # a batch takes the fast form's shapes only (d = 6, K <= 32, N >= 1024)
        BatchKMeans(n_clusters=3, sums="reference-order").fit(torch.randn(2, 5, 64, device=dev))
    with pytest.raises(ValueError):
        BatchKMeans(n_clusters=20, sums="fast")


def test_reference_order_batch_joint_stop_g7b(ops, oracle, dev):
    """BatchKMeans(sums="reference-order") on the reference's own l = 3 run (tests/golden/g7b): ONE loop for the batch, the
    error summed over the whole (l, d, K) tensor in ATen's order -- the reference's iteration count, labels, per-iteration
    errors and final centroid BITS; and bit for bit the oracle's restatement."""
    from eigentrajectory_amd import BatchKMeans
    z = G.load("g7b_batchkmeans_joint_stop.npz")
    km = BatchKMeans(n_clusters=int(z["K"]), n_redo=1, max_iter=100, tol=1e-4, init_mode="kmeans++", sums="reference-order")
    np.random.seed(0)
    labels = km.fit(T(z["x"], dev))
    assert km.n_iter_ == [len(z["trace"])] * 3
    assert np.array_equal(N_(labels).astype(np.uint8), z["labels"])
    assert np.array_equal(N_(km.centroids), z["centroids"])
    np.testing.assert_allclose(km.inertia_, z["trace"][-1, 1], rtol=1e-5)
    runs = ops.kmeans_fit_reference_order_batch(T(z["x"], dev), T(z["c0"], dev), 100, 1e-4)
    ref = oracle.kmeans_fit_batch_reference_order(list(z["x"]), list(z["c0"]), 100, 1e-4)
    for b, r in enumerate(runs):
        assert r["n_iter"] == ref["n_iter"]
        assert np.array_equal(N_(r["labels"]), ref["labels"][b])
        assert np.array_equal(N_(r["centroids"]), ref["centroids"][b])
        assert np.array_equal(N_(r["trace"])[:, 0], z["trace"][:, 0].astype(np.float32))
        np.testing.assert_allclose(r["inertia"], ref["inertia"][b], rtol=1e-5)


@pytest.mark.parametrize("filter_lp", [9, 4, -1])
@pytest.mark.parametrize("n,K,l", [(1024, 20, 2), (5003, 20, 3), (20001, 7, 4), (70000, 32, 2), (131072 + 13, 20, 2), (300000, 20, 2)])
def test_reference_order_fast_form_vs_oracle(ops, oracle, dev, et_option, n, K, l, filter_lp):
    """The one-launch-per-iteration form of the reference-order fit (csrc/et_kmeans_reforder.hip, namespace fast: parallel
    levels of ATen's cascade, permuted copy, last-arriver updates) against the oracle's literal restatement, on sizes that
    exercise every leftover of the cascade (partial chunk / group / block, N mod 4, N mod 32) and on batches: labels, centroid
    bits, per-iteration errors, iteration count; and problem 0 alone (l = 1: its own stop).  filter_lp = 4 switches the
    matrix-core label certification on (built, tested equal, off by default: HISTORY.md 3.8): the same bits; -1: the update kernel's grid
    form on the small shards that take the single-workgroup form by default."""
    from eigentrajectory_amd.synth import gaussian_points_np
    if filter_lp < 0:  # (-1: the update as a grid of block workgroups + last arriver also where one workgroup would do)
        et_option("reforder_single_update", 0)
    else:
        et_option("reforder_filter_min_lp", filter_lp)
    xs = np.stack([gaussian_points_np(6, n, seed=300 + 7 * b + n % 89, n_blobs=(0 if b % 2 else 5)) for b in range(l)])
    xs[0][:, ::61] *= np.float32(9.0)
    c0 = np.stack([oracle.kmeans_init_farthest(xs[b], K, (17 * (b + 1)) % n, reference_order=True)[0] for b in range(l)])
    iters = 12
    ref = oracle.kmeans_fit_batch_reference_order(list(xs), list(c0), iters, 1e-4)
    runs = ops.kmeans_fit_reference_order_batch(T(xs, dev), T(c0, dev), iters, 1e-4)
    for b, r in enumerate(runs):
        assert r["n_iter"] == ref["n_iter"], (b, r["n_iter"], ref["n_iter"])
        assert np.array_equal(N_(r["labels"]), ref["labels"][b]), b
        assert np.array_equal(N_(r["centroids"]), ref["centroids"][b], equal_nan=True), b
        assert np.array_equal(N_(r["trace"])[:, 0], ref["trace"][:, 0], equal_nan=True)
        np.testing.assert_allclose(r["inertia"], ref["inertia"][b], rtol=1e-5)
    one = oracle.kmeans_fit(xs[0], c0[0], iters, 1e-4, sums="reference-order")
    got = ops.kmeans_fit_reference_order(T(xs[0], dev), T(c0[0], dev), iters, 1e-4)
    assert got["n_iter"] == one["n_iter"] and np.array_equal(N_(got["labels"]), one["labels"])
    assert np.array_equal(N_(got["centroids"]), one["centroids"], equal_nan=True)
    assert np.array_equal(N_(got["trace"])[:, 0], one["trace"][:, 0], equal_nan=True)


def test_reference_order_fast_form_nan_centroids_and_huge_values(ops, oracle, dev):
    """An empty cluster (0/0 = NaN centroid, kmeans.py:182) and magnitudes near the fp32 range take the fast form's
    NaN-aware arg-max: still the oracle's bits (torch.max: a NaN beats everything, the first one stays)."""
    from eigentrajectory_amd.synth import gaussian_points_np
    x = gaussian_points_np(6, 6000, seed=5, n_blobs=4)
    c0 = x[:, :20].copy()
    c0[:, 7] = 1e6  # nobody's nearest centroid: empty after the first assignment -> NaN from the second iteration on
    ref = oracle.kmeans_fit(x, c0, 5, 1e-4, sums="reference-order")
    got = ops.kmeans_fit_reference_order(T(x, dev), T(c0, dev), 5, 1e-4)
    assert np.isnan(ref["centroids"]).any()
    assert got["n_iter"] == ref["n_iter"] and np.array_equal(N_(got["labels"]), ref["labels"])
    assert np.array_equal(N_(got["centroids"]), ref["centroids"], equal_nan=True)
    xb = (x * np.float32(3e18)).astype(np.float32)
    cb = xb[:, 100:120].copy()
    ref = oracle.kmeans_fit(xb, cb, 4, 1e-4, sums="reference-order")
    got = ops.kmeans_fit_reference_order(T(xb, dev), T(cb, dev), 4, 1e-4)
    assert np.array_equal(N_(got["labels"]), ref["labels"])
    assert np.array_equal(N_(got["centroids"]), ref["centroids"], equal_nan=True)


@pytest.mark.timeout(120)
@pytest.mark.parametrize("bad_problem", [0, 1])
def test_reference_order_batch_reports_bad_data_promptly(ops, dev, bad_problem):
    """NaN in ONE problem of a reference-order batch: every problem stops before the first iteration (the batch iterates
    jointly) and the call returns ValueError after one launch pair, not after max_iter of them."""
    import time
    from eigentrajectory_amd.synth import gaussian_points_np
    n = 20000
    x = np.stack([gaussian_points_np(6, n, seed=3 + b, n_blobs=4) for b in range(2)])
    c0 = np.ascontiguousarray(x[:, :, :20])
    ops.kmeans_fit_reference_order_batch(T(x, dev), T(c0, dev), 5, 1e-4, trace=False)  # (clean: warms the path up)
    x[bad_problem, 2, n // 2] = np.nan
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with pytest.raises(ValueError):
        ops.kmeans_fit_reference_order_batch(T(x, dev), T(c0, dev), 100000, 1e-4, trace=False)
    assert time.perf_counter() - t0 < 5.0
