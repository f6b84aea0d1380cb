"""GPU parity tests -- BatchKMeans, exact-sum path: farthest-first, Lloyd loop forms (exact scan, matrix-core filter, packed copy, chained, persistent, batch), the sharded drivers, the sklearn-recipe anchors.
HIP path (through the C ABI, via eigentrajectory_amd.ops) vs the CPU oracle and the golden vectors; needs a real MI355X:
run with ``pytest -m gpu``."""
import os

import numpy as np
import pytest
import torch

from . import _golden as G
from ._gpu_common import *  # noqa: F401,F403 -- fixtures (dev, ops) and helpers

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["gauss1000", "gauss10000", "blobs10000", "gauss100000", "ethm"])
def test_kmeans_bit_exact_vs_oracle_and_golden_g7(ops, oracle, dev, tag):
    z = G.load("g7_batchkmeans.npz")
    x = km_points(tag, z)
    first = int(z[f"{tag}.first_index"])
    c0 = ops.kmeans_init_farthest(T(x, dev), 20, first)
    assert np.array_equal(N_(c0), z[f"{tag}.c0"])  # the reference's 20 farthest-first picks, bit for bit
    res = fit_and_check_traceless(ops, T(x, dev), c0, 100, 1e-4)
    ref = oracle.kmeans_fit(x, z[f"{tag}.c0"], 100, 1e-4)
    assert res["n_iter"] == ref["n_iter"]
    assert np.array_equal(N_(res["labels"]), ref["labels"])            # bit-exact assignments
    assert np.array_equal(N_(res["centroids"]), ref["centroids"])      # bit-exact centroids
    assert np.array_equal(N_(res["trace"]), ref["trace"])
    # whole-run equality with the imported REFERENCE: the reference-order fit reproduces every case -- labels, iteration count,
    # centroid bits, the error of every iteration; the exact-sum fit (above: bit-exact against the oracle) every case but
    # gauss10000, where the ~1e-7 difference of the summation orders sends Lloyd to another fixed point (a rate)
    ro = ops.kmeans_fit_reference_order(T(x, dev), c0, 100, 1e-4)
    assert np.array_equal(N_(ro["labels"]), z[f"{tag}.labels"].astype(np.int64))
    assert ro["n_iter"] == len(z[f"{tag}.trace"])
    assert np.array_equal(N_(ro["centroids"]), z[f"{tag}.centroids"])
    assert np.array_equal(N_(ro["trace"])[:, 0], z[f"{tag}.trace"][:, 0].astype(np.float32))
    same = np.array_equal(N_(res["labels"]), z[f"{tag}.labels"].astype(np.int64)) and res["n_iter"] == len(z[f"{tag}.trace"])
    assert same == (tag != "gauss10000")
    # step-wise parity with the reference from ITS centroids (teacher forcing), first/last iterations
    hist = z[f"{tag}.history"]
    for i in (0, 1, len(hist) - 2):
        lb, _ = ops.kmeans_predict(T(x, dev), T(hist[i], dev))
        rl, _ = oracle.kmeans_assign(x, hist[i])
        assert np.array_equal(N_(lb), rl)
    assert np.array_equal(N_(lb), z[f"{tag}.labels"].astype(np.int64))


@pytest.mark.parametrize("n,d,K", [(1, 6, 1), (37, 6, 20), (1001, 6, 20), (4099, 3, 7), (5000, 2, 255), (2048, 16, 33),
                                   (999, 32, 5)])
def test_kmeans_shapes_vs_oracle(ops, oracle, dev, n, d, K):
    from eigentrajectory_amd.synth import gaussian_points_np
    K = min(K, n)
    x = gaussian_points_np(d, n, seed=n + d, n_blobs=5)
    c0 = ops.kmeans_init_farthest(T(x, dev), K, n // 2)
    r0, idx = oracle.kmeans_init_farthest(x, K, n // 2)
    assert np.array_equal(N_(c0), r0)
    res = fit_and_check_traceless(ops, T(x, dev), c0, 30, 1e-4)
    ref = oracle.kmeans_fit(x, r0, 30, 1e-4)
    assert res["n_iter"] == ref["n_iter"]
    assert np.array_equal(N_(res["labels"]), ref["labels"])
    assert np.array_equal(N_(res["centroids"]), ref["centroids"], equal_nan=True)
    np.testing.assert_array_equal(N_(res["trace"]), ref["trace"])
    lb, ms = ops.kmeans_predict(T(x, dev), res["centroids"])
    rl, rm = oracle.kmeans_assign(x, ref["centroids"])
    assert np.array_equal(N_(lb), rl) and np.array_equal(N_(ms), rm, equal_nan=True)
    np.testing.assert_array_equal(N_(ops.euc_sim(T(x[:, :50], dev), res["centroids"])),
                                  oracle.euc_sim(x[:, :50], ref["centroids"]))


def test_sharded_driver_on_one_gpu_matches_oracle(ops, oracle, dev):
    """dist.ShardedKMeans / fit_descriptor_sharded with the real device shard and no process group (world = 1): the
    step API, the candidate-selection kernel and the lagged convergence polling give the oracle's bits."""
    from eigentrajectory_amd.dist import ShardedKMeans, fit_descriptor_sharded
    from eigentrajectory_amd.synth import gaussian_points_np, synthetic_trajectories_np
    x = gaussian_points_np(6, 6000, seed=21, n_blobs=6)
    km = ShardedKMeans(T(x, dev), 20)
    c0 = km.init_farthest(1234)
    r0, _ = oracle.kmeans_init_farthest(x, 20, 1234)
    assert np.array_equal(N_(c0), r0)
    res = km.fit(c0.clone(), 60, 1e-4)
    ref = oracle.kmeans_fit(x, r0, 60, 1e-4)
    assert res["n_iter"] == ref["n_iter"]
    assert np.array_equal(N_(res["labels"]), ref["labels"])
    assert np.array_equal(N_(res["centroids"]), ref["centroids"])
    obs, pred = synthetic_trajectories_np(5000, seed=3)
    U_obs, U_pred, s_obs, s_pred, count = fit_descriptor_sharded(T(obs, dev), T(pred, dev), 6, ops.MODE_MOVING, 0.0, 1)
    g_obs, g_pred, _ = ops.fit_gram(T(obs, dev), T(pred, dev), ops.MODE_MOVING, 0.0, 1)
    assert count == 5000
    assert torch.equal(U_obs, ops.eigh_topk(g_obs, 6)[0]) and torch.equal(U_pred, ops.eigh_topk(g_pred, 6)[0])


@pytest.mark.parametrize("n,cut,trace", [(30000, 8192, False), (30000, 20004, True), (30000, 1024, False), (30000, 10001, False),
                                         (30000, 29501, True), (560000, 280000, False)])
def test_native_chained_loop_two_shards_one_gpu(ops, oracle, dev, n, cut, trace, et_option):
    """_two_shards_native exercises what a one-rank run cannot: the table being summed between the launches, the lockstep
    convergence polling, the final inertia reduction.  Centroids, labels, iteration count, error and inertia must be the
    oracle's on the whole data, bit for bit."""
    import ctypes as C
    from eigentrajectory_amd import _lib as L
    from eigentrajectory_amd.synth import gaussian_points_np
    # (the last case: both shards big enough for the PACKED copy of the points, each with its own origin and scale)
    et_option("kmeans_packed_min", 262144)
    K, max_iter, tol = 20, 40 if n <= 30000 else 16, 1e-4
    packed_fits = L.lib().et_internal_kmeans_packed_fits
    packed_fits.restype = C.c_longlong
    packed_before = packed_fits()
    x = gaussian_points_np(6, n, seed=31, n_blobs=9)
    x[:, ::53] *= 40.0
    c0, _ = oracle.kmeans_init_farthest(x, K, 77)
    ref = oracle.kmeans_fit(x, c0, max_iter, tol)
    cens, states, traces, labels = _two_shards_native(ops, dev, x, c0, cut, K, max_iter, tol, trace)
    assert packed_fits() == packed_before + (2 if n > 30000 and not trace else 0)
    for r in range(2):
        assert np.array_equal(N_(cens[r]), ref["centroids"]), r
        st = states[r]
        assert int(st.iter) == ref["n_iter"]
        assert np.float32(st.error) == np.float32(ref["error"]) and np.float32(st.inertia) == np.float32(ref["inertia"])
        if trace:
            assert np.array_equal(N_(traces[r])[:ref["n_iter"]], ref["trace"])
    assert np.array_equal(labels, ref["labels"])


@pytest.mark.parametrize("cut", [12000, 4100])
def test_native_two_shards_equal_the_rccl_world1_run(ops, dev, tmp_path, cut):
    """The two halves of what a multi-rank run is, on the SAME data: (i) et_kmeans_fit_sharded through a real RCCL
    communicator (world 1: one GPU per box here; ncclAllReduce is enqueued between the launches but has nobody to add),
    (ii) the same loop on two shards of this GPU with a test reduction where RCCL would add the ranks' tables.  Both end
    with the same centroids, labels and iteration count, bit for bit -- so the first N > 1 RCCL run has one unknown
    left, RCCL's own sum of 142 int64."""
    import socket
    import torch.multiprocessing as mp
    from eigentrajectory_amd.synth import gaussian_points_np
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_nccl_world1_worker, args=(port, str(tmp_path)), nprocs=1, join=True)
    r0 = np.load(tmp_path / "rank0.npz")  # (the worker asserts native == torch.distributed step API before it saves)
    x = gaussian_points_np(6, 24000, seed=6, n_blobs=7)  # the worker's data
    x[:, ::97] *= 300.0
    cens, states, _, labels = _two_shards_native(ops, dev, x, r0["c0"], cut, 20, 30, 1e-4, False)
    for r in range(2):
        assert np.array_equal(N_(cens[r]), r0["centroids"]) and int(states[r].iter) == int(r0["n_iter"])
    assert np.array_equal(labels, r0["labels"])


def test_sharded_path_over_rccl_world1(tmp_path, oracle):
    """The same sharded drivers with the "nccl" (= RCCL) backend initialised on the GPU (world size 1: one GPU per
    box here): every all-reduce / all-gather of the fit and of k-means goes through RCCL's device path."""
    import socket
    import torch.multiprocessing as mp
    from eigentrajectory_amd.synth import gaussian_points_np, synthetic_trajectories_np
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_nccl_world1_worker, args=(port, str(tmp_path)), nprocs=1, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    obs, pred = synthetic_trajectories_np(6000, seed=5)
    g_obs, g_pred, cnt = oracle.fit_gram(obs, pred, 2, 0.3, 1)
    assert int(r0["count"]) == cnt
    np.testing.assert_allclose(r0["U_pred"], oracle.eigh_topk(g_pred, 6)[0], atol=2e-6)
    x = gaussian_points_np(6, 24000, seed=6, n_blobs=7)
    x[:, ::97] *= 300.0
    c0, _ = oracle.kmeans_init_farthest(x, 20, 4321)
    assert np.array_equal(r0["c0"], c0)
    ref = oracle.kmeans_fit(x, c0, 30, 1e-4)
    assert int(r0["n_iter"]) == ref["n_iter"]
    assert np.array_equal(r0["centroids"], ref["centroids"]) and np.array_equal(r0["labels"], ref["labels"])


@pytest.mark.parametrize("cuts", [(0, 3000, 6000), (0, 257, 6000)])
def test_sharded_two_ranks_on_the_gpu(tmp_path, oracle, cuts):
    """Two processes, real device shards (both on cuda:0), gloo for the exchange: every rank ends with the oracle's
    single-process result -- the multi-GPU path minus RCCL."""
    import socket
    import torch.multiprocessing as mp
    from eigentrajectory_amd.synth import gaussian_points_np, synthetic_trajectories_np
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_two_rank_gpu_worker, args=(2, port, cuts, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    for key in ("U_pred", "count", "c0", "centroids", "n_iter"):
        assert np.array_equal(r0[key], r1[key], equal_nan=True), key
    obs, pred = synthetic_trajectories_np(6000, seed=5)
    g_obs, g_pred, cnt = oracle.fit_gram(obs, pred, 2, 0.3, 1)
    assert int(r0["count"]) == cnt
    np.testing.assert_allclose(r0["U_pred"], oracle.eigh_topk(g_pred, 6)[0], atol=2e-6)
    x = gaussian_points_np(6, 24000, seed=6, n_blobs=7)
    x[:, ::97] *= 300.0
    c0, _ = oracle.kmeans_init_farthest(x, 20, 4321)
    assert np.array_equal(r0["c0"], c0)
    ref = oracle.kmeans_fit(x, c0, 30, 1e-4)
    assert int(r0["n_iter"]) == ref["n_iter"]
    assert np.array_equal(r0["centroids"], ref["centroids"])
    assert np.array_equal(np.concatenate([r0["labels"], r1["labels"]]), ref["labels"])


@pytest.mark.parametrize("kind,n,K", [("blobs", 1024, 20), ("blobs", 4100, 3), ("blobs", 12288, 19), ("blobs", 8192, 21),
                                      ("blobs", 5000, 32), ("tiny", 4096, 20), ("huge", 4096, 20), ("outliers", 20000, 20),
                                      ("lattice", 6000, 20), ("lattice", 4096, 31), ("subnormal_mix", 7000, 20),
                                      ("line", 10000, 20), ("few_distinct", 4096, 20)])
def test_kmeans_filter_kernel_bit_exact_on_adversarial_data(ops, oracle, dev, kind, n, K):
    """Iterations >= 1 run the filter kernel (f16 MFMA upper bounds + exact certification): it may only ever say
    "label unchanged" when that is what the exact scan computes, whatever the data look like."""
    x = _filter_case(kind, n, seed=n + K)
    c0, _ = oracle.kmeans_init_farthest(x, K, n // 3)
    res = fit_and_check_traceless(ops, T(x, dev), T(c0, dev), 25, 1e-4)
    ref = oracle.kmeans_fit(x, c0, 25, 1e-4)
    assert res["n_iter"] == ref["n_iter"]
    assert np.array_equal(N_(res["labels"]), ref["labels"])
    assert np.array_equal(N_(res["centroids"]), ref["centroids"], equal_nan=True)
    np.testing.assert_array_equal(N_(res["trace"]), ref["trace"])


@pytest.mark.parametrize("tag", ["bench", "blobs", "offset", "outliers", "k3", "k32", "tiny", "huge", "lattice"])
def test_kmeans_packed_copy_equals_fp32_filter(ops, oracle, dev, tag, et_option):
    """Trace-less fits of big shards iterate on a packed copy of the points (f16 coordinates about a sample mean + a norm
    bound, 14 B per point; exact coordinates only for the points the test cannot decide).  Labels, centroids, iteration
    count, error and inertia must be those of the fp32 filter (and of the traced fit, which never uses the copy), bit for
    bit; the counter says which path ran."""
    import ctypes as C
    from eigentrajectory_amd import _lib as L
    fits = L.lib().et_internal_kmeans_packed_fits
    fits.restype = C.c_longlong
    et_option("kmeans_packed_min", 262144)  # (the library's own threshold is 2^17 points: where the copy pays)
    x, K, *rest = packed_case(tag, oracle)
    tol = rest[0] if rest else 1e-4
    x_dev = T(x, dev)
    c0 = ops.kmeans_init_farthest(x_dev, K, 4242 % x.shape[1])
    before = fits()
    res = fit_and_check_traceless(ops, x_dev, c0, 30, tol)  # traced fit == trace-less fit (packed)
    assert fits() == before + 1
    et_option("kmeans_packed", 0)
    plain = ops.kmeans_fit(x_dev, c0, 30, tol, trace=False)
    assert fits() == before + 1
    assert plain["n_iter"] == res["n_iter"] and torch.equal(plain["labels"], res["labels"])
    assert np.array_equal(N_(plain["centroids"]), N_(res["centroids"]), equal_nan=True)


@pytest.mark.parametrize("fused", ["1", "0"])
def test_kmeans_packed_copy_unusable_scale(ops, oracle, dev, et_option, fused):
    """magnitudes whose square leaves the fp32 range (the exact kernel decides every iteration): the packed copy reports
    itself unusable and the fit falls back -- the traced fit, the trace-less one and the oracle agree"""
    from eigentrajectory_amd.synth import gaussian_points_np
    et_option("kmeans_packed_min", 262144)
    et_option("kmeans_pack_fused", fused)
    x = gaussian_points_np(6, 262144, seed=33, n_blobs=6) * np.float32(1e24)
    x_dev = T(x, dev)
    c0 = ops.kmeans_init_farthest(x_dev, 20, 3)
    res = fit_and_check_traceless(ops, x_dev, c0, 6, 0.0)
    ref = oracle.kmeans_fit(x, N_(c0), 6, 0.0)
    assert res["n_iter"] == ref["n_iter"] and np.array_equal(N_(res["labels"]), ref["labels"])
    assert np.array_equal(N_(res["centroids"]), ref["centroids"], equal_nan=True)


def test_kmeans_packed_copy_written_by_its_own_pass(ops, oracle, dev, et_option):
    """option kmeans_pack_fused = 0: the copy is written by kmeans_pack_kernel before the loop instead of by the fit's first
    iteration -- same results"""
    et_option("kmeans_packed_min", 262144)
    x, K = packed_case("bench", oracle)[:2]
    x_dev = T(x, dev)
    c0 = ops.kmeans_init_farthest(x_dev, K, 17)
    fused = ops.kmeans_fit(x_dev, c0, 25, 1e-4, trace=False)
    et_option("kmeans_pack_fused", 0)
    own = ops.kmeans_fit(x_dev, c0, 25, 1e-4, trace=False)
    assert own["n_iter"] == fused["n_iter"] and torch.equal(own["labels"], fused["labels"])
    assert np.array_equal(N_(own["centroids"]), N_(fused["centroids"]), equal_nan=True)


@pytest.mark.parametrize("max_iter", [1, 2, 3, 7])
def test_kmeans_packed_copy_short_fits(ops, oracle, dev, et_option, max_iter):
    """the first launch of a fit is the exact scan, the packed body starts with the second: fits that end after one, two,
    three iterations, and one that converges before max_iter (well separated blobs), against the fp32 filter"""
    from eigentrajectory_amd.synth import gaussian_points_np
    et_option("kmeans_packed_min", 262144)
    x = gaussian_points_np(6, 262144 + 4 * 37, seed=21, n_blobs=20) * np.float32(1.0 if max_iter < 7 else 0.05)
    if max_iter == 7:  # 20 tight blobs, centres ~ N(0, 4^2): the farthest-first start is already near the fixed point
        x = x + gaussian_points_np(6, 1, seed=3)[:, :1] * 0.0
    x_dev = T(x, dev)
    c0 = ops.kmeans_init_farthest(x_dev, 20, 5)
    tol = 1e-4 if max_iter < 7 else 1e-2
    res = fit_and_check_traceless(ops, x_dev, c0, max_iter if max_iter < 7 else 60, tol)
    et_option("kmeans_packed", 0)
    plain = ops.kmeans_fit(x_dev, c0, max_iter if max_iter < 7 else 60, tol, trace=False)
    assert plain["n_iter"] == res["n_iter"] and plain["done"] == res["done"] and torch.equal(plain["labels"], res["labels"])
    assert np.array_equal(N_(plain["centroids"]), N_(res["centroids"]), equal_nan=True)


@pytest.mark.parametrize("extra", [4, 124, 128, 132, 252])
def test_kmeans_packed_copy_shard_tails(ops, oracle, dev, et_option, extra):
    """A pass of the packed body is 256 points: both lanes of a column request the column's rows of the lower AND of the
    upper 128-point block through buffer requests whose out-of-range offsets return zeros.  Shards whose last pass has one
    quad, an almost full lower block, no upper block, one quad of the upper block, all but one quad: against the oracle
    and the fp32 filter, bit for bit."""
    from eigentrajectory_amd.synth import gaussian_points_np
    et_option("kmeans_packed_min", 1024)
    et_option("kmeans_loop", "chain")
    n = 256 * 37 + extra
    x = gaussian_points_np(6, n, seed=100 + extra, n_blobs=9)
    x[:, n - 3:] *= np.float32(3.0)  # the shard's last points are ones whose labels move
    x_dev = T(x, dev)
    c0, _ = oracle.kmeans_init_farthest(x, 20, n - 1)
    ref = oracle.kmeans_fit(x, c0, 15, 1e-4)
    res = ops.kmeans_fit(x_dev, T(c0, dev), 15, 1e-4, trace=False)
    assert res["n_iter"] == ref["n_iter"]
    assert np.array_equal(N_(res["labels"]), ref["labels"])
    assert np.array_equal(N_(res["centroids"]), ref["centroids"])
    et_option("kmeans_packed", 0)
    plain = ops.kmeans_fit(x_dev, T(c0, dev), 15, 1e-4, trace=False)
    assert torch.equal(plain["labels"], res["labels"])


@pytest.mark.parametrize("copies", [1, 4, 8])
@pytest.mark.parametrize("K", [3, 20, 32])
def test_kmeans_chain_delta_table_copies(ops, oracle, dev, et_option, copies, K):
    """The chained loop of a single-GPU fit adds its per-iteration deltas onto several compact copies of the delta table
    (option kmeans_chain_copies, default 2; the host lowers the number until the copies fit the table: K = 3 has room for
    few).  Integer sums: every setting gives the bits of the oracle -- packed body and fp32 filter body, traced and not."""
    from eigentrajectory_amd.synth import gaussian_points_np
    et_option("kmeans_loop", "chain")
    et_option("kmeans_packed_min", 1024)
    et_option("kmeans_chain_copies", copies)
    n = 256 * 45 + 36
    x = gaussian_points_np(6, n, seed=300 + K, n_blobs=7)
    c0, _ = oracle.kmeans_init_farthest(x, K, 11)
    ref = oracle.kmeans_fit(x, c0, 12, 1e-4)
    for trace in (False, True):
        res = ops.kmeans_fit(T(x, dev), T(c0, dev), 12, 1e-4, trace=trace)
        assert res["n_iter"] == ref["n_iter"]
        assert np.array_equal(N_(res["labels"]), ref["labels"])
        assert np.array_equal(N_(res["centroids"]), ref["centroids"], equal_nan=True)


def test_kmeans_packed_copy_vs_oracle(ops, oracle, dev, et_option):
    """the packed path against the CPU oracle itself (one case: the oracle needs ~1 s per iteration at this size)"""
    import ctypes as C
    from eigentrajectory_amd import _lib as L
    et_option("kmeans_packed_min", 262144)
    fits = L.lib().et_internal_kmeans_packed_fits
    fits.restype = C.c_longlong
    before = fits()
    x, K = packed_case("blobs", oracle)[:2]
    c0, _ = oracle.kmeans_init_farthest(x, K, 99)
    ref = oracle.kmeans_fit(x, c0, 12, 1e-4)
    res = ops.kmeans_fit(T(x, dev), T(c0, dev), 12, 1e-4, trace=False)
    assert fits() == before + 1
    assert res["n_iter"] == ref["n_iter"]
    assert np.array_equal(N_(res["labels"]), ref["labels"])
    assert np.array_equal(N_(res["centroids"]), ref["centroids"])


def test_kmeans_fit_randomized_stress_vs_oracle(ops, oracle, dev):
    """40 seeded random configurations (size, K, scale over 16 decades, outliers, duplicated points, dead and
    near-collinear coordinates): farthest-first picks, labels, centroids and the error / inertia trace must be the
    oracle's bits in every one of them -- whichever of the kernels (exact scan, matrix-core filter, small-shard
    fused iteration, skipping farthest-first steps) the sizes select."""
    rng = np.random.default_rng(20240607)
    for case in range(40):
        n = int(rng.integers(256, 6000)) * 4
        K = int(rng.integers(3, 33))
        x = rng.standard_normal((6, n))
        nb = int(rng.integers(1, 12))
        x += rng.standard_normal((6, nb))[:, rng.integers(0, nb, size=n)] * rng.uniform(0.5, 8.0)
        if rng.random() < 0.4:   # heavy tail
            idx = rng.choice(n, max(1, n // int(rng.integers(20, 400))), replace=False)
            x[:, idx] *= 10.0 ** rng.uniform(1, 4)
        if rng.random() < 0.3:   # duplicated points
            src = rng.integers(0, n, size=n // 3)
            x[:, rng.integers(0, n, size=n // 3)] = x[:, src]
        if rng.random() < 0.3:   # a dead coordinate
            x[int(rng.integers(0, 6))] = 0.0
        if rng.random() < 0.3:   # two nearly collinear coordinates
            x[1] = x[0] * 1.5 + 1e-4 * rng.standard_normal(n)
        x = np.ascontiguousarray((x * 10.0 ** rng.uniform(-8, 8)).astype(np.float32))
        first = int(rng.integers(0, n))
        c0 = ops.kmeans_init_farthest(T(x, dev), K, first)
        r0, _ = oracle.kmeans_init_farthest(x, K, first)
        assert np.array_equal(N_(c0), r0, equal_nan=True), f"case {case}: farthest-first picks differ (n={n}, K={K})"
        res = fit_and_check_traceless(ops, T(x, dev), c0, 12, 1e-4)
        ref = oracle.kmeans_fit(x, r0, 12, 1e-4)
        assert res["n_iter"] == ref["n_iter"], f"case {case} (n={n}, K={K})"
        assert np.array_equal(N_(res["labels"]), ref["labels"]), f"case {case} (n={n}, K={K})"
        assert np.array_equal(N_(res["centroids"]), ref["centroids"], equal_nan=True), f"case {case} (n={n}, K={K})"
        np.testing.assert_array_equal(N_(res["trace"]), ref["trace"], err_msg=f"case {case} (n={n}, K={K})")


def test_kmeans_duplicates_nan_propagation_g7(ops, oracle, dev):
    from eigentrajectory_amd import BatchKMeans
    z = G.load("g7_batchkmeans.npz")
    x = z["dup.x"]
    c0 = z["dup.c0"]
    lb0, _ = ops.kmeans_predict(T(x, dev), T(c0, dev))
    assert np.array_equal(N_(lb0), z["dup.labels_iter0"])
    res = ops.kmeans_fit(T(x, dev), T(c0, dev), 5, 1e-4)
    ref = oracle.kmeans_fit(x, c0, 5, 1e-4)
    assert res["n_iter"] == 5 and np.isnan(res["inertia"]) and np.isnan(res["error"])
    assert np.array_equal(N_(res["centroids"]), ref["centroids"], equal_nan=True)
    assert np.array_equal(N_(res["labels"]), ref["labels"])
    km = BatchKMeans(n_clusters=20, max_iter=5)
    assert km.fit(T(x, dev)[None].contiguous(), T(c0, dev)[None]) is None  # like the reference (NaN inertia never wins)
    with pytest.raises(Exception):
        ops.kmeans_fit(T(np.full((6, 40), np.nan, np.float32), dev), T(c0, dev), 5, 1e-4)


def test_kmeans_sharded_steps_partition_independent(ops, oracle, dev):
    """The step API a multi-GPU run uses: shards of any size sum to the single-shard partials, bit for bit."""
    from eigentrajectory_amd.synth import gaussian_points_np
    x = gaussian_points_np(6, 10007, seed=4, n_blobs=9)
    c0, _ = oracle.kmeans_init_farthest(x, 20, 5)
    cuts = [0, 1, 4000, 4004, 10007]
    shards = [ops.KMeansShard(T(np.ascontiguousarray(x[:, a:b]), dev), 20) for a, b in zip(cuts[:-1], cuts[1:])]
    whole = ops.KMeansShard(T(x, dev), 20)
    for sh in shards + [whole]:
        sh.scan()
    mx = max(float(sh.state_f64[0].item()) for sh in shards)
    assert mx == float(whole.state_f64[0].item()) == float(np.abs(x).max())
    cen = [T(c0, dev).clone() for _ in shards]
    cw = T(c0, dev).clone()
    mn = min(int(sh.state[11].item()) for sh in shards)
    assert mn == int(whole.state[11].item()) == int(np.abs(x[x != 0]).min().view(np.uint32))
    for sh, c in zip(shards, cen):
        sh.state_f64[0] = mx  # what an all-reduce(MAX) leaves on every rank
        sh.state[11] = mn     # ... and the all-reduce(MIN) of the smallest non-zero magnitude
        sh.begin(10007, c)
    whole.begin(10007, cw)
    for it in range(6):
        total = sum(sh.assign(c).clone() for sh, c in zip(shards, cen))  # the all-reduce(SUM)
        pw = whole.assign(cw)
        assert torch.equal(total, pw)
        for sh, c in zip(shards, cen):
            sh.update(total, c, 1e-4)
        whole.update(pw, cw, 1e-4)
        assert all(torch.equal(c, cw) for c in cen)
    ref = oracle.kmeans_fit(x, c0, 6, 1e-4)
    assert np.array_equal(N_(cw), ref["centroids"])
    assert np.array_equal(np.concatenate([N_(sh.labels()) for sh in shards]), ref["labels"])


def test_sklearn_recipe_seeds_and_centres_vs_sklearn_g11(ops, dev):
    """The device recipe of anchor.py:65-71 against scikit-learn's own outputs (tests/golden/g11, captured by
    tools/make_golden_sklearn.py) and against the numpy restatement (oracle/sklearn_recipe.py)."""
    from eigentrajectory_amd.anchor import seeding_uniforms, sklearn_style_kmeans
    from eigentrajectory_amd.synth import gaussian_points_np
    from oracle import sklearn_recipe as R
    g11, g7 = G.load("g11_sklearn_anchors.npz"), G.load("g7_batchkmeans.npz")
    cases = {"ethm": g7["ethm.x"], "blobs20000": gaussian_points_np(6, 20000, seed=11, n_blobs=12)}
    for tag, C in cases.items():
        # pre-processing: numpy's float32 reduction order, bit for bit
        Xc, mean, tol = ops.center_columns(T(C, dev))
        r_x, r_mean, r_tol = R.center_columns(C)
        assert np.array_equal(N_(mean), r_mean) and np.array_equal(r_mean, g11[f"{tag}.mean"])
        assert np.array_equal(N_(Xc), r_x)
        assert np.float32(tol.item()) == r_tol == g11[f"{tag}.tol"]
        # seeding: the indices sklearn.cluster.kmeans_plusplus drew, ten initialisations on one stream
        U = seeding_uniforms(np.random.RandomState(0), 20, 10)
        assert np.array_equal(U, R.seeding_uniforms(np.random.RandomState(0), 20, 10))
        for i in range(10):
            c0, idx = ops.kmeanspp_seed(Xc, 20, torch.from_numpy(U[i]))
            assert np.array_equal(N_(idx), g11[f"{tag}.seeds"][i]), (tag, i, N_(idx), g11[f"{tag}.seeds"][i])
            assert np.array_equal(N_(c0), r_x[:, g11[f"{tag}.seeds"][i]])
        # the whole call: sklearn's centres (cluster order is sklearn's: same seeds -> same order)
        A, inertia, seeds = sklearn_style_kmeans(T(C, dev), 20)
        assert np.array_equal(N_(seeds), g11[f"{tag}.seeds"])
        ref = g11[f"{tag}.centers"]
        np.testing.assert_allclose(N_(A), ref, rtol=0, atol=5e-5 * np.abs(ref).max())
        assert abs(inertia * C.shape[1] / float(g11[f"{tag}.inertia"]) - 1) < 1e-5
        r = R.kmeans(C, 20)
        assert np.array_equal(N_(A), r["centers"])  # device recipe == numpy restatement, bit for bit
        # ten initialisations on ten streams / host threads or one after the other: the same result
        A_seq, inertia_seq, seeds_seq = sklearn_style_kmeans(T(C, dev), 20, concurrent=False)
        assert torch.equal(A_seq, A) and inertia_seq == inertia and torch.equal(seeds_seq, seeds)


def test_kmeanspp_seed_shapes_vs_oracle(ops, dev):
    """Seeding on other shapes (block boundaries of the running-sum search, tiny N, other d/K, duplicates)."""
    from eigentrajectory_amd.anchor import seeding_uniforms
    from eigentrajectory_amd.synth import gaussian_points_np
    from oracle import sklearn_recipe as R
    for n, d, K, seed in ((20, 6, 20, 1), (4096, 6, 20, 2), (4097, 3, 7, 3), (12289, 2, 33, 4), (100000, 6, 20, 5),
                          (70001, 16, 5, 6)):
        x = gaussian_points_np(d, n, seed=seed, n_blobs=5 if n > 100 else 0)
        if n == 4096:
            x[:, 100:200] = x[:, :1]  # duplicates: zero distances inside the running sum
        U = seeding_uniforms(np.random.RandomState(seed), K, 2)
        for i in range(2):
            c0, idx = ops.kmeanspp_seed(T(x, dev), K, torch.from_numpy(U[i]))
            r_c0, r_idx = R.kmeanspp_seed(x, K, U[i])
            assert np.array_equal(N_(idx), r_idx), (n, d, K, i, N_(idx), r_idx)
            assert np.array_equal(N_(c0), r_c0)


def test_batchkmeans_batch_of_problems_stops_together(ops, oracle, dev):
    """BatchKMeans.fit on (l, d, n) data (kmeans.py:200-259): the l problems run in lockstep and stop TOGETHER, on the
    error summed over the batch (kmeans.py:232, 239) -- bit for bit what the oracle's restatement of that loop gives."""
    from eigentrajectory_amd import BatchKMeans
    from eigentrajectory_amd.synth import gaussian_points_np
    xs = np.stack([gaussian_points_np(6, 3000, seed=60 + b, n_blobs=4 + b) for b in range(5)])
    km = BatchKMeans(n_clusters=12, max_iter=40)
    np.random.seed(3)
    labels = km.fit(T(xs, dev))
    assert labels.shape == (5, 3000) and km.centroids.shape == (5, 6, 12)
    np.random.seed(3)
    first = np.random.randint(3000)
    c0s = [oracle.kmeans_init_farthest(xs[b], 12, first)[0] for b in range(5)]
    refs = oracle.kmeans_fit_batch(list(xs), c0s, 40, 1e-4)
    alone = [oracle.kmeans_fit(xs[b], c0s[b], 40, 1e-4)["n_iter"] for b in range(5)]
    assert len(set(alone)) > 1  # the problems would stop at different iterations on their own
    for b in range(5):
        assert np.array_equal(N_(labels[b]), refs[b]["labels"]) and np.array_equal(N_(km.centroids[b]), refs[b]["centroids"])
        assert km.n_iter_[b] == refs[b]["n_iter"] == refs[0]["n_iter"]
    np.testing.assert_allclose(km.inertia_, np.mean([r["inertia"] for r in refs]), rtol=1e-6)


def test_batchkmeans_joint_stop_g7b(ops, dev):
    """The reference's own l = 3 run (tests/golden/g7b, tools/make_golden_batchkmeans.py): alone the problems take
    4 / 47 / 3 iterations, the batch takes 47 for all of them; labels equal, centroids to fp32 noise."""
    from eigentrajectory_amd import BatchKMeans
    z = G.load("g7b_batchkmeans_joint_stop.npz")
    km = BatchKMeans(n_clusters=int(z["K"]), n_redo=1, max_iter=100, tol=1e-4, init_mode="kmeans++")
    np.random.seed(0)
    labels = km.fit(T(z["x"], dev))
    assert km.n_iter_ == [len(z["trace"])] * 3
    assert np.array_equal(N_(labels).astype(np.uint8), z["labels"])
    np.testing.assert_allclose(N_(km.centroids), z["centroids"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(km.inertia_, z["trace"][-1, 1], rtol=1e-5)


def test_batchkmeans_helpers_run_the_batch_in_one_launch(ops, oracle, dev):
    """BatchKMeans.euc_sim / get_labels / predict on (l, d, n) operands (kmeans.py:59-76, 143-158): one launch for the
    batch, bit for bit what the per-problem oracle gives."""
    from eigentrajectory_amd import BatchKMeans
    from eigentrajectory_amd.synth import gaussian_points_np
    xs = np.stack([gaussian_points_np(6, 777, seed=90 + b, n_blobs=3 + b) for b in range(4)])
    cs = np.stack([xs[b][:, 5:300:23].copy() for b in range(4)])  # (4, 6, 13)
    sims = BatchKMeans.euc_sim(T(xs, dev), T(cs, dev))
    assert sims.shape == (4, 777, 13)
    km = BatchKMeans(n_clusters=13)
    maxsims, labels = km.get_labels(T(xs, dev), T(cs, dev))
    assert labels.shape == (4, 777) and labels.dtype == torch.int64
    for b in range(4):
        assert np.array_equal(N_(sims[b]), oracle.euc_sim(xs[b], cs[b]))
        lb, ms = oracle.kmeans_assign(xs[b], cs[b])
        assert np.array_equal(N_(labels[b]), lb) and np.array_equal(N_(maxsims[b]), ms)
    # leading dimensions beyond one (the reference's "...")
    sims2 = BatchKMeans.euc_sim(T(xs.reshape(2, 2, 6, 777), dev), T(cs.reshape(2, 2, 6, 13), dev))
    assert torch.equal(sims2.reshape(4, 777, 13), sims)


@pytest.mark.parametrize("n,K,B,shared", [(2048, 20, 10, True), (14456, 20, 10, True), (61896, 20, 10, True), (98304, 20, 10, True),
                                          (5000, 8, 3, False), (30000, 32, 4, False), (1000, 20, 5, True), (30002, 20, 3, True)])
def test_kmeans_fit_batch_equals_single_fits(ops, oracle, dev, n, K, B, shared):
    """et_kmeans_fit_batch (the problems as the y dimension of one persistent launch, in chunks when they do not fit on
    the device together; shapes it does not take run one after the other): every problem's centroids, iteration count,
    error, inertia and labels are bit for bit those of its own et_kmeans_fit -- i.e. the oracle's."""
    from eigentrajectory_amd.synth import gaussian_points_np
    xs = [gaussian_points_np(6, n, seed=70 + (0 if shared else b), n_blobs=6) for b in range(B)]
    for x in xs:
        x[:, ::89] *= 20.0
    rng = np.random.RandomState(n)
    c0 = np.stack([xs[b][:, rng.choice(n, K, replace=False)] for b in range(B)])
    X = T(xs[0], dev) if shared else T(np.stack(xs), dev)
    res = ops.kmeans_fit_batch(X, T(c0, dev), 60, 1e-4, want_labels=True)
    for b in range(B):
        one = ops.kmeans_fit(T(xs[b], dev), T(c0[b], dev), 60, 1e-4, trace=False)
        assert res["n_iter"][b] == one["n_iter"] and res["done"][b] == one["done"]
        # (an empty cluster is NaN in both, kmeans.py:182)
        assert np.array_equal(N_(res["centroids"][b]), N_(one["centroids"]), equal_nan=True)
        assert torch.equal(res["labels"][b], one["labels"])
        assert np.array_equal(np.float32([res["error"][b], res["inertia"][b]]), np.float32([one["error"], one["inertia"]]),
                              equal_nan=True)
    ref = oracle.kmeans_fit(xs[B - 1], c0[B - 1], 60, 1e-4)
    assert res["n_iter"][B - 1] == ref["n_iter"]
    assert np.array_equal(N_(res["centroids"][B - 1]), ref["centroids"], equal_nan=True)
    assert np.array_equal(N_(res["labels"][B - 1]), ref["labels"])
    no_labels = ops.kmeans_fit_batch(X, T(c0, dev), 60, 1e-4)
    assert no_labels["labels"] is None
    assert np.array_equal(N_(no_labels["centroids"]), N_(res["centroids"]), equal_nan=True)


@pytest.mark.parametrize("n,d,K", [(257, 6, 20), (4097, 6, 3), (20011, 6, 20), (70001, 4, 12), (5000, 9, 33)])
def test_kmeanspp_seed_batch_equals_single_seedings(ops, dev, n, d, K):
    """The batched seeding (seedings = the y dimension of every launch) draws, for every initialisation, the same seed
    indices and centres, bit for bit, as that initialisation alone -- which is pinned against scikit-learn's own
    kmeans_plusplus (G11)."""
    from eigentrajectory_amd.synth import gaussian_points_np
    import eigentrajectory_amd.anchor as A
    rng = np.random.RandomState(n + K)
    x = gaussian_points_np(d, n, seed=n % 97, n_blobs=7) if d == 6 else rng.standard_normal((d, n)).astype(np.float32)
    x[:, ::101] *= 30.0
    X = T(x, dev)
    U = torch.from_numpy(A.seeding_uniforms(np.random.RandomState(0), K, 10)).to(dev)
    cb, ib = ops.kmeanspp_seed_batch(X, K, U)
    for i in range(10):
        c1, i1 = ops.kmeanspp_seed(X, K, U[i])
        assert torch.equal(cb[i], c1) and torch.equal(ib[i], i1)
        assert np.array_equal(N_(c1), x[:, N_(i1)])


# ------------------------------------------------ et_kmeans_fit_batch: a problem whose grid barrier timed out
def test_kmeans_fit_batch_aborted_problem_is_refitted_from_its_initial_centroids(dev):
    """a problem of the side-by-side persistent launch whose barrier timed out never wrote its staged
    results; the collect step must leave the caller's initial centroids alone so that the chained refit starts from
    them.  The hook that marks problems as timed out after the launch exists only in libetamd_testhooks.so (the same
    sources with -DET_TEST_HOOKS; not in the product library), so this runs in a process of its own."""
    import subprocess
    import sys
    code = r"""
import ctypes, numpy as np, torch
from eigentrajectory_amd import ops, _lib as L
from eigentrajectory_amd.synth import gaussian_points_np
dev = torch.device("cuda:0")
n, K, B = 20000, 8, 5
x = gaussian_points_np(6, n, seed=71, n_blobs=8) * np.float32(3.0)
rng = np.random.RandomState(5)
c0 = np.stack([x[:, rng.choice(n, K, replace=False)] for _ in range(B)])
X, C0 = torch.from_numpy(x).to(dev), torch.from_numpy(c0).to(dev)
want = ops.kmeans_fit_batch(X, C0, 300, 1e-4, want_labels=True)
# (a refit that started from the converged centroids instead of the initial ones would stop after one or two iterations)
assert all(want["done"]) and min(want["n_iter"]) > 3
L.lib().et_testhook_kmeans_abort_mask(ctypes.c_ulonglong(0x0a))  # problems 1 and 3
got = ops.kmeans_fit_batch(X, C0, 300, 1e-4, want_labels=True)
assert got["n_iter"] == want["n_iter"] and got["done"] == want["done"]
assert torch.equal(got["centroids"], want["centroids"]) and torch.equal(got["labels"], want["labels"])
assert got["error"] == want["error"] and got["inertia"] == want["inertia"]
print("abort-refit ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ET_LIBETAMD=os.path.join(root, "eigentrajectory_amd", "libetamd_testhooks.so"), PYTHONPATH=root)
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=root, timeout=600)
    assert res.returncode == 0 and "abort-refit ok" in res.stdout, res.stdout + res.stderr


@pytest.mark.parametrize("n", [20000, 600000])
def test_kmeans_fit_batch_reports_bad_data(ops, dev, n):
    """NaN in the points: ValueError from the side-by-side path AND from the one-after-the-other fallback (shards that
    fill the device by themselves) -- ADVICE r3: the fallback used to return ET_OK."""
    from eigentrajectory_amd.synth import gaussian_points_np
    x = gaussian_points_np(6, n, seed=3, n_blobs=4)
    c0 = np.stack([x[:, :20], x[:, 20:40]])
    x[2, n // 2] = np.nan
    with pytest.raises(ValueError):
        ops.kmeans_fit_batch(T(x, dev), T(c0, dev), 10, 1e-4)
    with pytest.raises(ValueError):
        ops.kmeans_fit(T(x, dev), T(c0[0], dev), 10, 1e-4)
