"""Pin the CPU oracle against the golden vectors captured from the reference.

The reference has no tests; tests/golden/ holds inputs and the
reference's own outputs on them (tools/make_golden.py, fixtures G1-G9 of
SURVEY.md §8(c)).  Everything here runs on CPU.
"""
import numpy as np
import pytest

from . import _golden as G

FP = dict(rtol=1e-5, atol=1e-5)


def params_for(scene):
    g2 = G.load("g2_fit_all_scenes.npz")
    keys = ["ET_m_descriptor.U_obs_trunc", "ET_m_descriptor.U_pred_trunc", "ET_s_descriptor.U_obs_trunc",
            "ET_s_descriptor.U_pred_trunc", "ET_m_anchor.C_anchor", "ET_s_anchor.C_anchor"]
    return {k: g2[f"{scene}.{k}"] for k in keys}


def test_synth_generator_is_pinned():
    from eigentrajectory_amd.synth import gaussian_points_np, synthetic_trajectories_np
    z = G.load("synth_pin.npz")
    o, p = synthetic_trajectories_np(64, seed=0)
    assert np.array_equal(o, z["obs"]) and np.array_equal(p, z["pred"])
    assert np.array_equal(gaussian_points_np(6, 64, seed=11), z["pts"])


# ----------------------------------------------------------------------- G1
@pytest.mark.parametrize("sca", [True, False])
def test_g1_trajnorm(oracle, sca):
    g1 = G.load("g1_trajnorm_eth_test.npz")
    obs, pred, _ = G.dataset("eth", "test")
    t = "sca1" if sca else "sca0"
    ori, rot, s = oracle.norm_params(obs, sca)
    assert np.array_equal(ori, g1[t + "_ori"])
    np.testing.assert_allclose(rot, g1[t + "_rot"], atol=2e-7)
    fin = np.isfinite(g1[t + "_pred_norm"]).all(axis=(1, 2))
    if sca:
        ref = g1[t + "_sca"]
        assert np.array_equal(np.isfinite(s), np.isfinite(ref))  # motionless rows: inf in both
        m = np.isfinite(ref)
        np.testing.assert_allclose(s[m], ref[m], rtol=1e-6)
        assert (~fin).sum() == 30
    np.testing.assert_allclose(oracle.normalize(obs, obs, sca)[fin], g1[t + "_obs_norm"][fin], **FP)
    pn = oracle.normalize(obs, pred, sca)
    np.testing.assert_allclose(pn[fin], g1[t + "_pred_norm"][fin], **FP)
    np.testing.assert_allclose(oracle.denormalize(obs, pn, sca)[fin], g1[t + "_pred_roundtrip"][fin], **FP)


# ----------------------------------------------------------------------- G2
def test_fit_sets_reassemble_from_per_file_windows():
    """data/files/*.npz + splits.json (tools/make_golden_fitsets.py) reproduce the whole-split fixture of ETH."""
    for phase in ("train", "val"):
        for a, b in zip(G.dataset("eth", phase), G.dataset_from_files("eth", phase)):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("scene", G.SCENES)
def test_g2_fit_all_scenes(oracle, scene):
    """Gram + Jacobi on every split's own fit set (train+val+flip) vs the reference's SVD of the same rows."""
    g2 = G.load("g2_fit_all_scenes.npz")
    obs, pred = G.fit_input(scene)
    sd = G.static_dist(scene)
    flag = oracle.moving_flags(obs, sd)
    n_m, n_s = int(g2[f"{scene}.n_moving"]), int(g2[f"{scene}.n_static"])
    assert flag.sum() == n_m and (~flag).sum() == n_s
    if scene == "eth":
        assert (n_m, n_s) == (14456, 55860)
    for which, tag in ((1, "m"), (0, "s")):
        g_obs, g_pred, cnt = oracle.fit_gram(obs, pred, 2, sd, which)
        assert cnt == (n_m if which else n_s)
        assert np.allclose(g_obs, g_obs.T, rtol=1e-14) and np.allclose(g_pred, g_pred.T, rtol=1e-14)
        for name, gram, key in (("obs", g_obs, "U_obs_trunc"), ("pred", g_pred, "U_pred_trunc")):
            U, sigma = oracle.eigh_topk(gram, 6)
            U_ref = g2[f"{scene}.ET_{tag}_descriptor.{key}"]
            sig_ref = g2[f"{scene}.sigma_{name}_{tag}"]
            # LAPACK's signs are arbitrary -> align; its own fp32 error dominates the residual;
            # measured over the five scenes: <= 1.3e-5 (static obs, sigma_6/sigma_7 closest), typically 1e-6
            np.testing.assert_allclose(G.sign_align(U, U_ref), U_ref, atol=2e-5)
            np.testing.assert_allclose(U @ U.T, U_ref @ U_ref.T, atol=4e-5)
            np.testing.assert_allclose(sigma, sig_ref[:6], rtol=1e-5)
            np.testing.assert_allclose(U.T @ U, np.eye(6), atol=1e-6)
            assert (U[np.abs(U).argmax(axis=0), np.arange(6)] > 0).all()  # this build's sign convention


def test_eigh_matches_lapack_on_random_spd(oracle):
    rng = np.random.default_rng(5)
    for n in (1, 2, 7, 16, 24, 40):
        a = rng.standard_normal((n, n + 3))
        g = a @ a.T
        k = max(1, n // 2)
        U, sigma = oracle.eigh_topk(g, k)
        w, v = np.linalg.eigh(g)
        w, v = w[::-1][:k], v[:, ::-1][:, :k]
        np.testing.assert_allclose(sigma, np.sqrt(w), rtol=1e-6)
        np.testing.assert_allclose(G.sign_align(U, v), v, atol=1e-5)


# ----------------------------------------------------------------------- G3
@pytest.mark.parametrize("scene", G.SCENES)
def test_g3_descriptor_evaluation_table(oracle, scene):
    """script/descriptor_evaluation.py:87-112, TrajNorm(sca=False), k = 1..12."""
    g3 = G.load("g3_descriptor_evaluation.npz")
    obs, pred, _ = G.dataset(scene, "test")
    g_obs, g_pred, cnt = oracle.fit_gram(obs, pred, 0, 0.0, 0)
    assert cnt == obs.shape[0]
    errs = np.zeros((12, 2))
    for k in range(1, 13):
        Uo, so = oracle.eigh_topk(g_obs, k)
        Up, sp = oracle.eigh_topk(g_pred, k)
        c_obs, c_pred, _, _ = oracle.norm_project(obs, pred, None, None, Uo, Up, 0)
        ro = oracle.anchor_reconstruct(c_obs[:, :, None], obs, None, None, None, Uo, 0)[0]
        rp = oracle.anchor_reconstruct(c_pred[:, :, None], obs, None, None, None, Up, 0)[0]
        errs[k - 1] = [np.linalg.norm(ro - obs, axis=-1).mean(), np.linalg.norm(rp - pred, axis=-1).mean()]
    np.testing.assert_allclose(errs, g3[f"{scene}.err"], atol=1e-4)
    np.testing.assert_allclose(so[:12], g3[f"{scene}.sigma_obs"][:12], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(sp, g3[f"{scene}.sigma_pred"][:12], rtol=2e-4, atol=2e-4)


# ------------------------------------------------------------------- G4 / G5
@pytest.mark.parametrize("tag,mode", [("m", 1), ("s", 0)])
def test_g4_g5_project_reconstruct(oracle, tag, mode):
    z = G.load("g45_project_reconstruct_eth_test.npz")
    p = params_for("eth")
    obs, pred, _ = G.dataset("eth", "test")
    rows = z[f"{tag}.rows"]
    obs, pred = obs[rows], pred[rows]
    Uo, Up, A = p[f"ET_{tag}_descriptor.U_obs_trunc"], p[f"ET_{tag}_descriptor.U_pred_trunc"], p[f"ET_{tag}_anchor.C_anchor"]
    c_obs, c_pred, nrm, flag = oracle.norm_project(obs, pred, Uo, Up, Uo, Up, mode)
    np.testing.assert_allclose(c_obs, z[f"{tag}.C_obs"], **FP)
    np.testing.assert_allclose(c_pred, z[f"{tag}.C_pred"], **FP)
    assert np.array_equal(nrm[:2].T, obs[:, -1]) and (flag == mode).all()
    recon = oracle.anchor_reconstruct(z[f"{tag}.C_refine"], obs, A, A, Up, Up, mode)
    np.testing.assert_allclose(recon, z[f"{tag}.recon"], rtol=1e-5, atol=2e-5)
    dC = oracle.anchor_reconstruct_bwd(z[f"{tag}.dtraj"], obs, Up, Up, mode)
    np.testing.assert_allclose(dC, z[f"{tag}.dC"], rtol=1e-5, atol=2e-5)


# ----------------------------------------------------------------------- G6
@pytest.mark.parametrize("scene", G.SCENES)
@pytest.mark.parametrize("stub", ["zero", "linear"])
def test_g6_wrapper_ade_fde(oracle, scene, stub):
    """Same weights + same inputs => same ADE/FDE (|delta| <= 1e-5)."""
    from oracle import wrapper_ref as W
    g6 = G.load("g6_wrapper_stub_predictors.npz")
    p = params_for(scene)
    obs, pred, sse = G.dataset(scene, "test")
    predictor = W.zero_stub(6, 20) if stub == "zero" else W.linear_stub(g6["linear_stub_w"])
    sd = G.static_dist(scene)
    if scene == "univ":
        sse = sse[::8]  # keep the CPU suite short; the GPU tests run every scene
    ades, fdes, ref_a, ref_f, losses = [], [], [], [], []
    for i, (s, e) in enumerate(sse):
        out = W.forward(p, obs[s:e], pred[s:e], predictor, sd)
        ades.append(W.batch_ade(out["recon_traj"], pred[s:e]))
        fdes.append(W.batch_fde(out["recon_traj"], pred[s:e]))
        ref_a.append(g6[f"{scene}.{stub}.ade"][s:e])
        ref_f.append(g6[f"{scene}.{stub}.fde"][s:e])
        losses.append([out["loss_eigentraj"], out["loss_euclidean_ade"], out["loss_euclidean_fde"]])
    ades, fdes, ref_a, ref_f = map(np.concatenate, (ades, fdes, ref_a, ref_f))
    np.testing.assert_allclose(ades, ref_a, atol=1e-5)
    np.testing.assert_allclose(fdes, ref_f, atol=1e-5)
    assert abs(ades.mean() - ref_a.mean()) < 1e-5 and abs(fdes.mean() - ref_f.mean()) < 1e-5
    ref_l = g6[f"{scene}.{stub}.losses"][:: (8 if scene == "univ" else 1)]
    np.testing.assert_allclose(np.asarray(losses), ref_l, rtol=1e-5, atol=1e-5)
    if scene == "eth":
        np.testing.assert_allclose(out["recon_traj"], g6[f"eth.{stub}.recon_last"], rtol=1e-5, atol=2e-5)


# ------------------------------------------------------------------- G7 / G8
def test_g8_euc_sim_bits(oracle):
    z = G.load("g8_euc_sim.npz")
    y = oracle.euc_sim(z["a"], z["b"])
    # the dot product (fmaf chain) and |a|^2 are bit-identical to torch; torch's reduction of a
    # 20-column |b|^2 uses a tail path with another summation tree (1 ulp on 7 columns)
    f32 = np.float32
    an = np.zeros(96, f32)
    bn = np.zeros(20, f32)
    for i in range(6):
        an = (an + z["a"][i] * z["a"][i]).astype(f32)
        bn = (bn + z["b"][i] * z["b"][i]).astype(f32)
    assert np.array_equal(an, z["a_norm"])
    same_cols = bn == z["b_norm"]
    assert same_cols.sum() >= 13
    assert np.array_equal(y[:, same_cols], z["y"][:, same_cols])
    np.testing.assert_allclose(y, z["y"], atol=4e-6)
    # rebuilding y from torch's own norms reproduces every bit -> the operation order is the reference's
    dot = ((y + bn[None, :]) + an[:, None])  # not exact inverse; check via direct recomputation instead
    del dot
    acc = np.zeros((96, 20), np.float64)
    d32 = np.zeros((96, 20), f32)
    for i in range(6):
        d32 = (z["a"][i].astype(np.float64)[:, None] * z["b"][i].astype(np.float64)[None, :] + d32).astype(f32)
    del acc
    y2 = ((d32 * f32(2) - z["a_norm"][:, None]).astype(f32) - z["b_norm"][None, :]).astype(f32)
    assert np.array_equal(y2, z["y"])


G7_CASES = ["gauss1000", "gauss10000", "blobs10000", "gauss100000", "ethm"]
# Whole-run label equality of the EXACT-sum fit with the reference is ill-conditioned: the reference's fp32 per-cluster
# sums carry ~1e-7 rounding noise that Lloyd iterations amplify chaotically (gauss10000 ends in another local optimum;
# measured rate over 96 runs: test_g7c_whole_runs_vs_reference).  Step-wise parity (below) holds for every iteration of
# every case, and the reference-order mode reproduces gauss10000 as well (test_g7_whole_run_reference_order).
G7_WHOLE_RUN_EQUAL = ["gauss1000", "blobs10000", "gauss100000", "ethm"]


def g7_points(z, tag):
    from eigentrajectory_amd.synth import gaussian_points_np
    if tag == "ethm":
        return z["ethm.x"]
    n = int(tag.replace("gauss", "").replace("blobs", ""))
    return gaussian_points_np(6, n, seed=11, n_blobs=int(z[f"{tag}.blobs"]))


@pytest.mark.parametrize("tag", G7_CASES)
def test_g7_farthest_first_init(oracle, tag):
    z = G.load("g7_batchkmeans.npz")
    x = g7_points(z, tag)
    first = int(z[f"{tag}.first_index"])
    c0, idx = oracle.kmeans_init_farthest(x, 20, first)
    assert idx[0] == first and len(set(idx.tolist())) == 20
    assert np.array_equal(c0, z[f"{tag}.c0"])  # the same 20 data points, bit for bit
    assert np.array_equal(c0, x[:, idx])


@pytest.mark.parametrize("tag", G7_CASES)
def test_g7_lloyd_step_parity(oracle, tag):
    """Teacher-forced: from the reference's centroids entering iteration i, one oracle step must give
    the reference's labels (bit-exact through the next centroids' support) and centroids i+1."""
    z = G.load("g7_batchkmeans.npz")
    x = g7_points(z, tag)
    hist, trace = z[f"{tag}.history"], z[f"{tag}.trace"]
    n = x.shape[1]
    frac = oracle.kmeans_frac_bits(float(np.abs(x).max()), n)
    assert len(hist) == len(trace) + 1
    steps = range(len(trace)) if n <= 20000 else list(range(0, len(trace), 7)) + [len(trace) - 1]
    for i in steps:
        sf = oracle.kmeans_sim_frac_bits(float(np.abs(x).max()), float(np.abs(hist[i]).max()), 6, n)
        labels, sums, counts, ss, nn = oracle.kmeans_assign_accumulate(x, hist[i], frac, sf)
        c_new, err, ine, done = oracle.kmeans_update(sums, counts, ss, nn, n, frac, sf, 1e-4, hist[i])
        # the reference sums fp32 in torch's order: a few ulp on the mean
        np.testing.assert_allclose(c_new, hist[i + 1], rtol=5e-7, atol=1e-6)
        np.testing.assert_allclose(ine, trace[i, 1], rtol=5e-6)
        np.testing.assert_allclose(err, trace[i, 0], rtol=1e-3, atol=1e-7)
        assert done == (i == len(trace) - 1 and len(trace) < 100)
    assert np.array_equal(labels, z[f"{tag}.labels"].astype(np.int64))  # last assignment == returned labels


@pytest.mark.parametrize("tag", G7_WHOLE_RUN_EQUAL)
def test_g7_batchkmeans_whole_run(oracle, tag):
    from eigentrajectory_amd.synth import gaussian_points_np
    z = G.load("g7_batchkmeans.npz")
    x = g7_points(z, tag)
    res = oracle.kmeans_fit(x, z[f"{tag}.c0"], 100, 1e-4)
    ref_trace = z[f"{tag}.trace"]
    assert res["n_iter"] == len(ref_trace)
    assert np.array_equal(res["labels"], z[f"{tag}.labels"].astype(np.int64))
    np.testing.assert_allclose(res["centroids"], z[f"{tag}.centroids"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(res["trace"][-1, 1], ref_trace[-1, 1], rtol=1e-5)
    if f"{tag}.query_labels" in z:
        q = gaussian_points_np(6, 512, seed=12, n_blobs=int(z[f"{tag}.blobs"]))
        ql, _ = oracle.kmeans_assign(q, res["centroids"])
        assert (ql != z[f"{tag}.query_labels"]).sum() <= 1  # centroids differ in the last ulp


# ---- G7c: whole runs of the imported reference over many seeds; the reference-order mode ----
def g7c_case(z, n, seed):
    from eigentrajectory_amd.synth import gaussian_points_np
    tag = f"n{n}.s{seed}"
    return tag, gaussian_points_np(6, n, seed=seed, n_blobs=int(z[f"{tag}.blobs"]))


def g7c_same(z, tag, labels, n_iter):
    import hashlib
    return (hashlib.sha256(labels.astype(np.uint8).tobytes()).digest() == bytes(z[f"{tag}.labels_sha256"])
            and n_iter == int(z[f"{tag}.n_iter"]))


def test_reforder_arithmetic_equals_torch(oracle):
    """The third-party arithmetic the reference-order mode restates (ATen's CPU sum kernel; torch is not under
    /root/reference) pinned against torch itself, bit for bit: kmeans.py:59-76 euc_sim, :180-182 masked sums, :50 error."""
    import torch
    torch.set_num_threads(1)
    rng = np.random.default_rng(3)
    f32 = np.float32
    for d in (6, 1, 2, 5, 9, 16, 17, 20, 32):
        for m, n in [(96, 20), (1031, 19), (33, 33), (64, 65), (5, 1), (7, 3), (1, 1), (12, 8), (40003, 20)]:
            if d != 6 and m > 2000:
                continue
            a = (rng.standard_normal((d, m)) * 3).astype(f32)
            b = (rng.standard_normal((d, n)) * 3).astype(f32)
            at, bt = torch.from_numpy(a), torch.from_numpy(b)
            y = at.transpose(-2, -1) @ bt  # kmeans.py:71-74, the reference's own statements
            y.mul_(2)
            y.sub_(at.pow(2).sum(dim=-2)[..., :, None])
            y.sub_(bt.pow(2).sum(dim=-2)[..., None, :])
            assert np.array_equal(oracle.euc_sim(a, b, reference_order=True), y.numpy()), (d, m, n)
    assert not np.array_equal(oracle.euc_sim(a, b), y.numpy())  # the build's own order differs in the last bit somewhere
    for n in (1, 3, 5, 17, 63, 64, 65, 1000, 1023, 1025, 4097, 65537, 100000, 300001):
        x = (rng.standard_normal((6, n)) * 3 + 1).astype(f32)
        lab = rng.integers(0, 20, size=n)
        xt, lt = torch.from_numpy(x)[None], torch.from_numpy(lab)[None]
        mask = torch.stack([lt == i for i in range(20)], dim=-1)  # kmeans.py:180-182
        want = (xt.unsqueeze(dim=-1) * mask.unsqueeze(dim=-3)).sum(dim=-2)[0].numpy()
        assert np.array_equal(oracle.kmeans_reforder_sums(x, lab, 20), want), n
    for size in (1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 31, 33, 120, 121, 127, 128, 255, 256, 1000, 6 * 255):
        v = (rng.standard_normal(size) ** 2).astype(f32)
        assert oracle.inner_sum(v) == torch.from_numpy(v).sum().item(), size


@pytest.mark.parametrize("n", [1000, 10000])
def test_g7c_whole_runs_vs_reference(oracle, n):
    """BatchKMeans whole runs (farthest-first seeding + <= 100 Lloyd iterations) of the imported reference on 32 data
    sets per size.  reference-order sums: every run ends with the reference's labels, iteration count and centroid
    BITS.  Exact sums (the default): the same initial centroids everywhere; whole-run equality is a rate -- 32/32 at
    N = 1e3, 31/32 at 1e4, 13/32 at 1e5 (tools/g7c_rate.py prints the whole table; N = 1e5 is sampled below)."""
    z = G.load("g7c_batchkmeans_seeds.npz")
    equal_exact = 0
    for seed in z["seeds"]:
        tag, x = g7c_case(z, n, int(seed))
        first = int(z[f"{tag}.first_index"])
        c0, _ = oracle.kmeans_init_farthest(x, 20, first, reference_order=True)
        assert np.array_equal(c0, z[f"{tag}.c0"])
        assert np.array_equal(oracle.kmeans_init_farthest(x, 20, first)[0], c0)  # the build's own order picks the same points
        r = oracle.kmeans_fit(x, c0, 100, 1e-4, sums="reference-order")
        assert g7c_same(z, tag, r["labels"], r["n_iter"]), tag
        assert np.array_equal(r["labels"], z[f"{tag}.labels"])
        assert np.array_equal(r["centroids"], z[f"{tag}.centroids"]), tag
        assert r["trace"][-1, 0] == np.float32(z[f"{tag}.final_error_inertia"][0])  # torch's error, every bit
        np.testing.assert_allclose(r["trace"][-1, 1], z[f"{tag}.final_error_inertia"][1], rtol=1e-5)
        e = oracle.kmeans_fit(x, c0, 100, 1e-4)
        equal_exact += g7c_same(z, tag, e["labels"], e["n_iter"])
    assert equal_exact == {1000: 32, 10000: 31}[n]


@pytest.mark.parametrize("seed,exact_equal", [(101, True), (112, False), (117, False), (130, True)])
def test_g7c_whole_runs_vs_reference_1e5(oracle, seed, exact_equal):
    """... and a sample of the N = 1e5 runs (all 32: tools/g7c_rate.py, and on the GPU test_reference_order_kmeans_g7c)."""
    z = G.load("g7c_batchkmeans_seeds.npz")
    tag, x = g7c_case(z, 100000, seed)
    c0, _ = oracle.kmeans_init_farthest(x, 20, int(z[f"{tag}.first_index"]), reference_order=True)
    assert np.array_equal(c0, z[f"{tag}.c0"])
    r = oracle.kmeans_fit(x, c0, 100, 1e-4, sums="reference-order")
    assert g7c_same(z, tag, r["labels"], r["n_iter"])
    assert np.array_equal(r["centroids"], z[f"{tag}.centroids"])
    e = oracle.kmeans_fit(x, c0, 100, 1e-4)
    assert g7c_same(z, tag, e["labels"], e["n_iter"]) == exact_equal


@pytest.mark.parametrize("tag", G7_CASES)
def test_g7_whole_run_reference_order(oracle, tag):
    """All five G7 runs -- gauss10000 included -- end with the reference's labels in the reference-order mode."""
    z = G.load("g7_batchkmeans.npz")
    x = g7_points(z, tag)
    res = oracle.kmeans_fit(x, z[f"{tag}.c0"], 100, 1e-4, sums="reference-order")
    assert res["n_iter"] == len(z[f"{tag}.trace"])
    assert np.array_equal(res["labels"], z[f"{tag}.labels"].astype(np.int64))
    assert np.array_equal(res["centroids"], z[f"{tag}.centroids"])
    assert np.array_equal(res["trace"][:, 0], z[f"{tag}.trace"][:, 0].astype(np.float32))  # torch's error, every iteration


def test_g7b_batch_of_problems_stops_on_the_summed_error(oracle):
    """kmeans.py:228-240 with l = 3 problems (fixture from the imported reference, tools/make_golden_batchkmeans.py):
    alone they stop after 4 / 47 / 3 iterations, together after 47 -- the error is ONE sum over the batch."""
    z = G.load("g7b_batchkmeans_joint_stop.npz")
    x, c0, K = z["x"], z["c0"], int(z["K"])
    assert list(z["iterations_alone"]) != [len(z["trace"])] * 3  # the fixture does separate the two semantics
    for b in range(3):
        r0, _ = oracle.kmeans_init_farthest(x[b], K, int(z["first_index"]))
        assert np.array_equal(r0, c0[b])
    runs = oracle.kmeans_fit_batch(list(x), list(c0), 100, 1e-4)
    assert [r["n_iter"] for r in runs] == [len(z["trace"])] * 3
    for b in range(3):
        assert np.array_equal(runs[b]["labels"], z["labels"][b].astype(np.int64))
        np.testing.assert_allclose(runs[b]["centroids"], z["centroids"][b], rtol=2e-5, atol=2e-5)
    # the reference's printed error of iteration t = the sum over the problems; its inertia = the mean over the batch
    err = sum(r["trace"][:, 0].astype(np.float64) for r in runs)
    np.testing.assert_allclose(err, z["trace"][:, 0], rtol=2e-3, atol=1e-7)
    ine = np.mean([r["trace"][:, 1].astype(np.float64) for r in runs], axis=0)
    np.testing.assert_allclose(ine, z["trace"][:, 1], rtol=1e-5)


def test_g7b_batch_in_reference_order_reproduces_the_reference(oracle):
    """The same l = 3 run in the reference's summation orders (kmeans.py:228-240 with ONE error over the contiguous
    (l, d, K) tensor in ATen's inner-sum order): iteration count, labels, the printed error of EVERY iteration and the final
    centroid BITS of the imported reference."""
    z = G.load("g7b_batchkmeans_joint_stop.npz")
    x, c0, K = z["x"], z["c0"], int(z["K"])
    for b in range(3):
        r0, _ = oracle.kmeans_init_farthest(x[b], K, int(z["first_index"]), reference_order=True)
        assert np.array_equal(r0, c0[b])
    r = oracle.kmeans_fit_batch_reference_order(list(x), list(c0), 100, 1e-4)
    assert r["n_iter"] == len(z["trace"])
    assert np.array_equal(r["labels"], z["labels"].astype(np.int64))
    assert np.array_equal(r["centroids"], z["centroids"])
    assert np.array_equal(r["trace"][:, 0], z["trace"][:, 0].astype(np.float32))
    np.testing.assert_allclose(r["trace"][:, 1], z["trace"][:, 1], rtol=1e-5)


def test_g7_duplicate_points_nan_propagation(oracle):
    """kmeans.py:180-182: an empty cluster becomes NaN and poisons every later label (no re-seeding)."""
    z = G.load("g7_batchkmeans.npz")
    x = z["dup.x"]  # 30 points, 8 distinct: farthest-first must pick duplicates (ties decided by fp noise)
    c0, idx = oracle.kmeans_init_farthest(x, 20, int(z["dup.first_index"]))
    assert np.array_equal(c0, x[:, idx])
    assert len({tuple(c) for c in c0[:, :8].T}) == 8  # the 8 distinct points come first, as in the reference
    assert np.array_equal(c0[:, :8], z["dup.c0"][:, :8])
    c0 = z["dup.c0"]  # continue from the reference's picks
    lb0, _ = oracle.kmeans_assign(x, c0)
    assert np.array_equal(lb0, z["dup.labels_iter0"])
    res = oracle.kmeans_fit(x, c0, 1, 1e-4)
    assert np.array_equal(np.isnan(res["centroids"]), np.isnan(z["dup.centroids_iter0"]))
    m = ~np.isnan(res["centroids"])
    np.testing.assert_allclose(res["centroids"][m], z["dup.centroids_iter0"][m], rtol=1e-6)
    lb1, ms1 = oracle.kmeans_assign(x, res["centroids"])
    assert np.array_equal(lb1, z["dup.labels_iter1"]) and np.isnan(ms1).all() and z["dup.maxsims_iter1_isnan"].all()
    res5 = oracle.kmeans_fit(x, c0, 5, 1e-4)
    assert res5["n_iter"] == 5 and np.isnan(res5["inertia"]) and np.isnan(res5["error"])
    assert bool(z["dup.fit_returned_none"])  # the reference returns None here (inertia NaN never < 1e32)


def test_kmeans_partials_are_partition_independent(oracle):
    """Exact int64 partial sums: any sharding of the points gives identical bits."""
    from eigentrajectory_amd.synth import gaussian_points_np
    x = gaussian_points_np(6, 5000, seed=3, n_blobs=7)
    c0, _ = oracle.kmeans_init_farthest(x, 20, 17)
    frac = oracle.kmeans_frac_bits(float(np.abs(x).max()), x.shape[1])
    sfrac = oracle.kmeans_sim_frac_bits(float(np.abs(x).max()), float(np.abs(c0).max()), 6, x.shape[1])
    _, s_all, c_all, ss_all, _ = oracle.kmeans_assign_accumulate(x, c0, frac, sfrac)
    for cuts in ([2500], [1, 4999], [1000, 1001, 3000, 4096]):
        s, c, ss = np.zeros_like(s_all), np.zeros_like(c_all), 0
        for part in np.split(np.arange(5000), cuts):
            xs = np.ascontiguousarray(x[:, part])
            _, ps, pc, pss, _ = oracle.kmeans_assign_accumulate(xs, c0, frac, sfrac)
            s += ps
            c += pc
            ss += pss
        assert np.array_equal(s, s_all) and np.array_equal(c, c_all) and ss == ss_all


# ----------------------------------------------------------------------- G9
def test_g9_ade_fde():
    from oracle import wrapper_ref as W
    z = G.load("g9_metrics.npz")
    np.testing.assert_allclose(W.batch_ade(z["pred"], z["gt"]), z["ade"], rtol=1e-6)
    np.testing.assert_allclose(W.batch_fde(z["pred"], z["gt"]), z["fde"], rtol=1e-6)


# ---------------------------------------------------------------------- G11
@pytest.mark.parametrize("tag", ["ethm", "blobs20000"])
def test_g11_sklearn_recipe_restatement(oracle, tag):
    """oracle/sklearn_recipe.py against scikit-learn's own outputs (tools/make_golden_sklearn.py): the pre-processing
    bit for bit, the k-means++ seed indices of ten initialisations on one RandomState(0) stream, and the final
    centres / inertia of KMeans(n_clusters=20, random_state=0, init='k-means++', n_init=10) (anchor.py:65-71)."""
    from eigentrajectory_amd.synth import gaussian_points_np
    from oracle import sklearn_recipe as R
    g11 = G.load("g11_sklearn_anchors.npz")
    C = G.load("g7_batchkmeans.npz")["ethm.x"] if tag == "ethm" else gaussian_points_np(6, 20000, seed=11, n_blobs=12)
    r = R.kmeans(C, 20)
    assert np.array_equal(r["mean"], g11[f"{tag}.mean"]) and r["tol"] == g11[f"{tag}.tol"]
    assert np.array_equal(r["seeds"], g11[f"{tag}.seeds"])
    ref = g11[f"{tag}.centers"]
    np.testing.assert_allclose(r["centers"], ref, rtol=0, atol=5e-5 * np.abs(ref).max())  # same order: same seeds
    assert abs(r["inertia"] / float(g11[f"{tag}.inertia"]) - 1) < 1e-5


def _same_distinct_centres(mine, ref, tol):
    """the two centre sets cover each other: every column of one has a column of the other within tol"""
    mine, ref = np.asarray(mine, np.float64), np.asarray(ref, np.float64)
    dist = np.linalg.norm(mine[:, :, None] - ref[:, None, :], axis=0)
    return bool(dist.min(axis=1).max() <= tol and dist.min(axis=0).max() <= tol)


def test_g11_empty_clusters_are_relocated_like_sklearn(oracle):
    """anchor.py:65-71 on an input with more clusters than distinct points (15 locations x 8 copies, K = 20): sklearn
    re-seeds the 5 clusters that stay empty with points far from their centres and ends with the 15 locations + 5
    duplicates, inertia 0 -- so does the restated loop.  WHICH locations are duplicated is not comparable: all candidate
    distances are rounding noise of sklearn's own float32 sums (~1e-14), and its seeding of the last five centres samples
    from distances that are zero up to the rounding of a BLAS dgemm.)"""
    from oracle import sklearn_recipe as R
    g11 = G.load("g11_sklearn_anchors.npz")
    C, ref = g11["dup15.x"], g11["dup15.centers"]
    assert len({tuple(c) for c in C.T}) == 15 and len(set(g11["dup15.labels"].tolist())) == 15
    X, mean, tol = R.center_columns(C)
    c0, _ = R.kmeanspp_seed(X, 20, R.seeding_uniforms(np.random.RandomState(0), 20, 1)[0])
    assert not np.isfinite(oracle.kmeans_fit(X, c0, 300, float(tol))["centroids"]).all()  # BatchKMeans semantics: NaN
    r = R.kmeans(C, 20)
    assert np.isfinite(r["centers"]).all()
    assert _same_distinct_centres(r["centers"], ref, 1e-5)
    assert len({tuple(np.round(c, 4)) for c in r["centers"].T}) == 15 == len({tuple(np.round(c, 4)) for c in ref.T})
    assert r["inertia"] < 1e-3 and float(g11["dup15.inertia"]) < 1e-8  # (fp32 cancellation noise of 2ab - |a|^2 - |b|^2, 120 points)


def test_g11_recipe_is_the_reference_anchor_fit_eth(oracle):
    """The recipe on the ETH moving coefficients reproduces the anchors the REFERENCE's calculate_parameters stored
    (G2 `eth.ET_m_anchor.C_anchor`, anchor.py:65-74), cluster for cluster."""
    from oracle import sklearn_recipe as R
    r = R.kmeans(G.load("g7_batchkmeans.npz")["ethm.x"], 20)
    ref = G.load("g2_fit_all_scenes.npz")["eth.ET_m_anchor.C_anchor"]
    np.testing.assert_allclose(r["centers"], ref, rtol=0, atol=5e-5 * np.abs(ref).max())
