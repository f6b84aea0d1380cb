"""GPU parity tests -- TrajNorm, projection, reconstruction (+ fused metrics), the descriptor fit and BASELINE.json's own sizes.
HIP path (through the C ABI, via eigentrajectory_amd.ops) vs the CPU oracle and the golden vectors; needs a real MI355X:
run with ``pytest -m gpu``."""
import os

import numpy as np
import pytest
import torch

from . import _golden as G
from ._gpu_common import *  # noqa: F401,F403 -- fixtures (dev, ops) and helpers

pytestmark = pytest.mark.gpu


# --------------------------------------------------------------------------------- TrajNorm
@pytest.mark.parametrize("sca", [True, False])
def test_trajnorm_vs_oracle_and_golden(ops, oracle, dev, sca):
    from eigentrajectory_amd import TrajNorm
    g1 = G.load("g1_trajnorm_eth_test.npz")
    obs, pred, _ = G.dataset("eth", "test")
    tn = TrajNorm(ori=True, rot=True, sca=sca)
    tn.calculate_params(T(obs, dev))
    t = "sca1" if sca else "sca0"
    assert np.array_equal(N_(tn.traj_ori), g1[t + "_ori"])
    np.testing.assert_allclose(N_(tn.traj_rot), g1[t + "_rot"], atol=5e-7)
    fin = np.isfinite(g1[t + "_pred_norm"]).all(axis=(1, 2))
    pn = tn.normalize(T(pred, dev))
    np.testing.assert_allclose(N_(pn)[fin], g1[t + "_pred_norm"][fin], **FP)
    np.testing.assert_allclose(N_(pn)[fin], oracle.normalize(obs, pred, sca)[fin], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(N_(tn.denormalize(pn))[fin], pred[fin], rtol=1e-5, atol=1e-5)
    if sca:
        assert np.array_equal(np.isfinite(N_(tn.traj_sca)), np.isfinite(g1[t + "_sca"]))
    # flag subsets (normalizer.py:20-28 are independent switches)
    tn2 = TrajNorm(ori=True, rot=False, sca=False)
    tn2.calculate_params(T(obs, dev))
    assert tn2.traj_rot is None and tn2.traj_sca is None
    np.testing.assert_array_equal(N_(tn2.normalize(T(pred, dev))), pred - obs[:, -1:, :])


# ------------------------------------------------------------------------------- projection
@pytest.mark.parametrize("n", [0, 1, 255, 256, 257, 1000, 2305, 3500, 70001])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_project_fast_path_vs_oracle(ops, oracle, dev, n, mode):
    p = eth_params()
    obs, pred = synth(max(n, 1), seed=3, min_disp=1e-3 if mode == 1 else 0.0)
    obs, pred = obs[:n], pred[:n]
    us = [p["ET_m_descriptor.U_obs_trunc"], p["ET_m_descriptor.U_pred_trunc"], p["ET_s_descriptor.U_obs_trunc"],
          p["ET_s_descriptor.U_pred_trunc"]]
    c_obs, c_pred, nrm, flag = ops.norm_project(T(obs, dev), T(pred, dev), *(T(u, dev) for u in us), mode, 0.3)
    assert c_obs.shape == (6, n) and c_pred.shape == (6, n) and nrm.shape == (4, n)
    if n == 0:
        return
    r_obs, r_pred, r_nrm, r_flag = oracle.norm_project(obs, pred, *us, mode, 0.3)
    assert np.array_equal(N_(flag), r_flag) and np.array_equal(N_(nrm), r_nrm)
    close(N_(c_obs), r_obs)
    close(N_(c_pred), r_pred)
    # obs-only (inference) form
    c_obs2, c_none, _, _ = ops.norm_project(T(obs, dev), None, T(us[0], dev), None, T(us[2], dev), None, mode, 0.3)
    assert c_none is None and torch.equal(c_obs2, c_obs)


@pytest.mark.parametrize("k,t_obs,t_pred", [(1, 8, 12), (4, 8, 12), (12, 8, 12), (3, 5, 7), (16, 8, 12), (6, 3, 1)])
def test_project_reconstruct_generic_dims_vs_oracle(ops, oracle, dev, k, t_obs, t_pred):
    from eigentrajectory_amd.synth import synthetic_trajectories_np
    rng = np.random.default_rng(k * 100 + t_obs)
    obs, pred = synthetic_trajectories_np(777, seed=5, obs_len=t_obs, pred_len=t_pred, min_disp=1e-3)
    us = [rng.standard_normal((2 * t, k)).astype(np.float32) for t in (t_obs, t_pred, t_obs, t_pred)]
    for mode in (0, 1, 2):
        c_obs, c_pred, nrm, flag = ops.norm_project(T(obs, dev), T(pred, dev), *(T(u, dev) for u in us), mode, 0.3)
        r_obs, r_pred, r_nrm, r_flag = oracle.norm_project(obs, pred, *us, mode, 0.3)
        assert np.array_equal(N_(flag), r_flag)
        close(N_(c_obs), r_obs)
        close(N_(c_pred), r_pred)
        s = 3
        cr = rng.standard_normal((k, 777, s)).astype(np.float32)
        a_m, a_s = rng.standard_normal((k, s)).astype(np.float32), rng.standard_normal((k, s)).astype(np.float32)
        rec = ops.anchor_reconstruct(T(cr, dev), T(a_m, dev), T(a_s, dev), T(us[1], dev), T(us[3], dev), mode, 0.3,
                                     obs=T(obs, dev))
        ref = oracle.anchor_reconstruct(cr, obs, a_m, a_s, us[1], us[3], mode, 0.3)
        close(N_(rec), ref)
        dt = rng.standard_normal(ref.shape).astype(np.float32)
        from eigentrajectory_amd.ops import _reconstruct_bwd
        dC = _reconstruct_bwd(T(dt, dev), T(obs, dev), None, T(us[1], dev), T(us[3], dev), mode, 0.3, t_obs)
        close(N_(dC), oracle.anchor_reconstruct_bwd(dt, obs, us[1], us[3], mode, 0.3))


def test_projection_golden_g4(ops, dev):
    z = G.load("g45_project_reconstruct_eth_test.npz")
    p = eth_params()
    obs, pred, _ = G.dataset("eth", "test")
    for tag, mode in (("m", 1), ("s", 0)):
        rows = z[f"{tag}.rows"]
        uo, up = p[f"ET_{tag}_descriptor.U_obs_trunc"], p[f"ET_{tag}_descriptor.U_pred_trunc"]
        c_obs, c_pred, _, _ = ops.norm_project(T(obs[rows], dev), T(pred[rows], dev), T(uo, dev), T(up, dev), T(uo, dev),
                                               T(up, dev), mode)
        np.testing.assert_allclose(N_(c_obs), z[f"{tag}.C_obs"], **FP)
        np.testing.assert_allclose(N_(c_pred), z[f"{tag}.C_pred"], **FP)


# --------------------------------------------------------------------------- reconstruction
@pytest.mark.parametrize("s", [1, 2, 20, 37, 256, 300])
@pytest.mark.parametrize("n", [1, 13, 256, 1001])
def test_reconstruct_fwd_bwd_vs_oracle(ops, oracle, dev, s, n):
    p = eth_params()
    rng = np.random.default_rng(s * 1000 + n)
    obs, _ = synth(n, seed=9)
    um, us_ = p["ET_m_descriptor.U_pred_trunc"], p["ET_s_descriptor.U_pred_trunc"]
    a_m = rng.standard_normal((6, s)).astype(np.float32)
    a_s = rng.standard_normal((6, s)).astype(np.float32)
    c = rng.standard_normal((6, n, s)).astype(np.float32)
    _, _, nrm, _ = ops.norm_project(T(obs, dev), None, T(p["ET_m_descriptor.U_obs_trunc"], dev), None,
                                    T(p["ET_s_descriptor.U_obs_trunc"], dev), None, 2, 0.3)
    ct = T(c, dev).requires_grad_(True)
    rec = ops.anchor_reconstruct(ct, T(a_m, dev), T(a_s, dev), T(um, dev), T(us_, dev), 2, 0.3, nrm=nrm)
    ref = oracle.anchor_reconstruct(c, obs, a_m, a_s, um, us_, 2, 0.3)
    assert rec.shape == (s, n, 12, 2)
    close(N_(rec), ref)
    rec_obs = ops.anchor_reconstruct(T(c, dev), T(a_m, dev), T(a_s, dev), T(um, dev), T(us_, dev), 2, 0.3, obs=T(obs, dev))
    assert torch.equal(rec_obs, rec.detach())  # cached nrm and obs give the same normaliser state
    dt = rng.standard_normal(ref.shape).astype(np.float32)
    (rec * T(dt, dev)).sum().backward()
    close(N_(ct.grad), oracle.anchor_reconstruct_bwd(dt, obs, um, us_, 2, 0.3))


def test_reconstruction_golden_g5(ops, dev):
    z = G.load("g45_project_reconstruct_eth_test.npz")
    p = eth_params()
    obs, _, _ = G.dataset("eth", "test")
    for tag, mode in (("m", 1), ("s", 0)):
        rows = z[f"{tag}.rows"]
        up, a = p[f"ET_{tag}_descriptor.U_pred_trunc"], p[f"ET_{tag}_anchor.C_anchor"]
        ct = T(z[f"{tag}.C_refine"], dev).requires_grad_(True)
        rec = ops.anchor_reconstruct(ct, T(a, dev), T(a, dev), T(up, dev), T(up, dev), mode, obs=T(obs[rows], dev))
        np.testing.assert_allclose(N_(rec), z[f"{tag}.recon"], rtol=1e-5, atol=3e-5)
        (rec * T(z[f"{tag}.dtraj"], dev)).sum().backward()
        np.testing.assert_allclose(N_(ct.grad), z[f"{tag}.dC"], rtol=1e-5, atol=3e-5)


# -------------------------------------------------------------------------------------- fit
def test_fit_gram_and_eigh_vs_oracle_and_golden_g2(ops, oracle, dev):
    g2 = G.load("g2_fit_all_scenes.npz")
    obs, pred = G.eth_fit_input()
    sd = G.static_dist("eth")
    for which, tag in ((1, "m"), (0, "s")):
        g_obs, g_pred, cnt = ops.fit_gram(T(obs, dev), T(pred, dev), 2, sd, which)
        r_obs, r_pred, r_cnt = oracle.fit_gram(obs, pred, 2, sd, which)
        assert int(cnt.item()) == r_cnt == int(g2[f"eth.n_{'moving' if which else 'static'}"])
        # the fp32 normalised rows differ from the oracle's in the last ulp (sincosf/atan2f of the
        # device library vs glibc), which bounds the agreement of the sums; the exact-summation
        # check is test_fit_gram_summation_exact below
        for g, r in ((g_obs, r_obs), (g_pred, r_pred)):
            g = N_(g)
            assert np.array_equal(g, g.T)
            close(g, r, tol=1e-6)
        for name, g, key in (("obs", g_obs, "U_obs_trunc"), ("pred", g_pred, "U_pred_trunc")):
            U, sigma = ops.eigh_topk(g, 6)
            Ur, sr = oracle.eigh_topk(N_(g), 6)  # same matrix in -> the oracle's Jacobi to fp32 rounding
            np.testing.assert_allclose(N_(U), Ur, rtol=0, atol=1e-6)
            np.testing.assert_allclose(N_(sigma), sr, rtol=1e-6)
            U_ref = g2[f"eth.ET_{tag}_descriptor.{key}"]
            np.testing.assert_allclose(G.sign_align(N_(U), U_ref), U_ref, atol=2e-5)
            np.testing.assert_allclose(N_(sigma), g2[f"eth.sigma_{name}_{tag}"][:6], rtol=1e-5)


@pytest.mark.parametrize("n,t_obs,t_pred", [(100000, 8, 12), (12345, 8, 12), (5000, 5, 7), (1, 8, 12), (63, 8, 12), (65, 8, 12), (257, 8, 12)])
def test_fit_gram_summation_exact(ops, dev, n, t_obs, t_pred):
    """Identity mode takes the rows as they are, so the only arithmetic is sum_n x_i x_j: products of
    fp32 values are exact in fp64 and the fp64 sums must agree with numpy's to ~1e-13."""
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, t_obs, 2)).astype(np.float32)
    b = rng.standard_normal((n, t_pred, 2)).astype(np.float32) * 7
    g_obs, g_pred, cnt = ops.fit_gram(T(a, dev), T(b, dev), ops.MODE_IDENTITY, which=0)
    assert int(cnt.item()) == n
    for g, x in ((g_obs, a), (g_pred, b)):
        m = x.reshape(n, -1).astype(np.float64)
        ref = m.T @ m
        np.testing.assert_allclose(N_(g), ref, rtol=0, atol=1e-12 * np.abs(ref).max())


@pytest.mark.parametrize("n", [1, 3, 63, 64, 65, 127, 255, 256, 257, 1000, 49153])
def test_fit_gram_ragged_sizes_vs_oracle(ops, oracle, dev, n):
    """The (8, 12) Gram kernel takes 64 trajectories per wavefront pass: sizes around the pass / workgroup edges, every
    normalising mode (identity: test_fit_gram_summation_exact), against the oracle (normalised rows agree to the last ulp or two, hence the 1e-6) and the row counts exactly."""
    rng = np.random.default_rng(n)
    obs = np.cumsum(rng.standard_normal((n, 8, 2)).astype(np.float32) * 0.4 + 0.3, axis=1).astype(np.float32)
    pred = (obs[:, -1:] + np.cumsum(rng.standard_normal((n, 12, 2)).astype(np.float32) * 0.4 + 0.3, axis=1)).astype(np.float32)
    obs[::7, -3] = obs[::7, -1]  # motionless rows: static in SPLIT mode, identity rotation
    for mode, which in ((ops.MODE_MOVING, 1), (ops.MODE_STATIC, 0), (ops.MODE_SPLIT, 1), (ops.MODE_SPLIT, 0)):
        o, p = obs, pred
        if mode == ops.MODE_MOVING:  # the scale normalisation divides by the last displacement: no motionless rows here
            o = obs.copy()
            o[::7, -3] = o[::7, -1] - 0.25
        g_obs, g_pred, cnt = ops.fit_gram(T(o, dev), T(p, dev), mode, 0.2, which)
        r_obs, r_pred, r_cnt = oracle.fit_gram(o, p, mode, 0.2, which)
        assert int(cnt.item()) == r_cnt, (mode, which)
        for g, r in ((g_obs, r_obs), (g_pred, r_pred)):
            g = N_(g)
            assert np.array_equal(g, g.T)
            close(g, r, tol=1e-6)


def test_eigh_vs_oracle(ops, oracle, dev):
    """The Jacobi kernel against the oracle's Jacobi (same pairing, sweeps and update order; the kernel takes the
    square roots of the rotation parameters from v_rsq_f64 + two Goldschmidt steps where the oracle's are correctly
    rounded): U to ~1e-14 in fp64, i.e. the same fp32 value except on a rounding boundary; a solve is deterministic and
    the batched launch gives the single launch's bits."""
    rng = np.random.default_rng(1)
    differing = total = 0
    for n in (1, 2, 5, 16, 24, 33, 64):
        a = rng.standard_normal((n, n + 2))
        g = a @ a.T
        k = max(1, n // 2)
        U, s = ops.eigh_topk(T(g, dev), k)
        Ur, sr = oracle.eigh_topk(g, k)
        (Ub, sb), (Ub2, _) = ops.eigh_topk_batch([T(g, dev), T(g[:8, :8].copy(), dev)], [k, min(k, 8)])  # one launch
        assert torch.equal(Ub, U) and torch.equal(sb, s)
        U2, s2 = ops.eigh_topk(T(g, dev), k)
        assert torch.equal(U2, U) and torch.equal(s2, s)
        np.testing.assert_allclose(N_(Ub2), oracle.eigh_topk(np.ascontiguousarray(g[:8, :8]), min(k, 8))[0], rtol=0, atol=2e-7)
        np.testing.assert_allclose(N_(U), Ur, rtol=0, atol=2e-7)  # |U| <= 1: an ulp of fp32 is <= 6e-8
        np.testing.assert_allclose(N_(s), sr, rtol=3e-7)
        differing += int((N_(U) != Ur).sum())
        total += Ur.size
    assert differing <= max(1, total // 200), f"{differing} of {total} fp32 entries of U differ from the oracle's"


def test_fit_generic_dims_and_truncated_svd(ops, oracle, dev):
    from eigentrajectory_amd import ETDescriptor
    from eigentrajectory_amd.synth import synthetic_trajectories_np
    from eigentrajectory_amd.utils import default_hyper_params
    obs, pred = synthetic_trajectories_np(3000, seed=2, obs_len=5, pred_len=7, min_disp=1e-3)
    for which in (1, 0):
        g_obs, g_pred, cnt = ops.fit_gram(T(obs, dev), T(pred, dev), 2, 0.3, which)
        r_obs, r_pred, r_cnt = oracle.fit_gram(obs, pred, 2, 0.3, which)
        assert int(cnt.item()) == r_cnt
        close(N_(g_obs), r_obs, tol=1e-6)
        close(N_(g_pred), r_pred, tol=1e-6)
    # truncated_SVD API (descriptor.py:91-114) against torch's SVD
    d = ETDescriptor(default_hyper_params(obs_len=5, pred_len=7, k=4))
    xn = torch.from_numpy(oracle.normalize(obs, pred, True))
    U, S, V = d.truncated_SVD(xn.to(dev))
    Ur, Sr, Vtr = torch.linalg.svd(xn.reshape(-1, 14).T.double(), full_matrices=False)
    np.testing.assert_allclose(N_(S), Sr[:4].numpy(), rtol=1e-5)
    np.testing.assert_allclose(G.sign_align(N_(U), Ur[:, :4].numpy()), Ur[:, :4].numpy(), atol=2e-5)
    M = xn.reshape(-1, 14).T.numpy()
    np.testing.assert_allclose((N_(U) * N_(S)) @ N_(V).T, Ur[:, :4].numpy() * Sr[:4].numpy() @ Vtr[:4].numpy(), atol=2e-3)
    assert M.shape == (14, 3000)


@pytest.mark.parametrize("scene", G.SCENES)
def test_descriptor_evaluation_table_g3(dev, scene):
    """config 1 of BASELINE.json: script/descriptor_evaluation.py's SVD table, k = 1..12, to the printed 4 decimals."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("descriptor_evaluation", os.path.join(os.path.dirname(G.GOLDEN), "..", "scripts",
                                                                                        "descriptor_evaluation.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g3 = G.load("g3_descriptor_evaluation.npz")
    obs, pred, _ = G.dataset(scene, "test")
    table = mod.svd_table(T(obs, dev), T(pred, dev))
    np.testing.assert_allclose(table, g3[f"{scene}.err"], atol=1e-4)


@pytest.mark.parametrize("s,n,k,t_pred", [(20, 181, 6, 12), (1, 1000, 6, 12), (256, 5, 6, 12), (300, 7, 6, 12), (3, 50, 4, 7)])
def test_fused_metrics_epilogue_vs_oracle(ops, oracle, dev, s, n, k, t_pred):
    from oracle import wrapper_ref as W
    from eigentrajectory_amd.synth import synthetic_trajectories_np
    rng = np.random.default_rng(s + n)
    obs, gt = synthetic_trajectories_np(n, seed=8, pred_len=t_pred)
    um, us_ = (rng.standard_normal((2 * t_pred, k)).astype(np.float32) * 0.3 for _ in range(2))
    a_m, a_s = (rng.standard_normal((k, s)).astype(np.float32) for _ in range(2))
    c = rng.standard_normal((k, n, s)).astype(np.float32)
    ade, fde = ops.anchor_reconstruct_metrics(T(c, dev), T(gt, dev), T(a_m, dev), T(a_s, dev), T(um, dev), T(us_, dev), 2,
                                              0.3, obs=T(obs, dev))
    rec = oracle.anchor_reconstruct(c, obs, a_m, a_s, um, us_, 2, 0.3)
    close(N_(ade), W.batch_ade(rec, gt), tol=2e-6)
    close(N_(fde), W.batch_fde(rec, gt), tol=2e-6)


@pytest.mark.parametrize("variant", ["0", "f32", "1"])
@pytest.mark.parametrize("s,mode", [(20, 2), (20, 1), (12, 2), (33, 0), (64, 2)])
def test_fused_metrics_kernel_variants_vs_oracle(ops, oracle, dev, et_option, variant, s, mode):
    """The three forms of the S >= 12 epilogue -- vector-ALU tile kernel (option metrics_form = t), fp32 matrix instructions
    (f), two-term f16 matrix instructions (default) -- against the oracle's reconstruction + compute_batch_ade / fde
    (utils/metrics.py), per-row descriptor choice included; n is not a multiple of the 64 / S rows of a pass."""
    from oracle import wrapper_ref as W
    from eigentrajectory_amd.synth import synthetic_trajectories_np
    et_option("metrics_form", {"0": "t", "f32": "f", "1": "a"}[variant])
    n = 1003
    rng = np.random.default_rng(s + mode)
    obs, gt = synthetic_trajectories_np(n, seed=9)
    um, us_ = (rng.standard_normal((24, 6)).astype(np.float32) * 0.3 for _ in range(2))
    a_m, a_s = (rng.standard_normal((6, s)).astype(np.float32) for _ in range(2))
    c = rng.standard_normal((6, n, s)).astype(np.float32)
    ade, fde = ops.anchor_reconstruct_metrics(T(c, dev), T(gt, dev), T(a_m, dev), T(a_s, dev), T(um, dev), T(us_, dev), mode,
                                              0.3, obs=T(obs, dev))
    rec = oracle.anchor_reconstruct(c, obs, a_m, a_s, um, us_, mode, 0.3)
    close(N_(ade), W.batch_ade(rec, gt), tol=2e-6)
    close(N_(fde), W.batch_fde(rec, gt), tol=2e-6)


@pytest.mark.parametrize("n,s", [(3, 20), (4, 20), (5, 20), (7, 20), (1, 64), (2, 33), (5, 12), (6, 12), (6145 * 3 + 1, 20)])
def test_fused_metrics_matrix_kernel_few_rows_and_tail_passes(ops, oracle, dev, n, s):
    """The persistent matrix-core kernel at its edges: one pass only, a last pass moved back over its predecessor's rows,
    fewer passes than wavefronts, and one pass more than a whole round of the resident wavefronts."""
    from oracle import wrapper_ref as W
    from eigentrajectory_amd.synth import synthetic_trajectories_np
    rng = np.random.default_rng(n + s)
    obs, gt = synthetic_trajectories_np(n, seed=11)
    um, us_ = (rng.standard_normal((24, 6)).astype(np.float32) * 0.3 for _ in range(2))
    a_m, a_s = (rng.standard_normal((6, s)).astype(np.float32) for _ in range(2))
    c = rng.standard_normal((6, n, s)).astype(np.float32)
    ade, fde = ops.anchor_reconstruct_metrics(T(c, dev), T(gt, dev), T(a_m, dev), T(a_s, dev), T(um, dev), T(us_, dev), 2,
                                              0.3, obs=T(obs, dev))
    rec = oracle.anchor_reconstruct(c, obs, a_m, a_s, um, us_, 2, 0.3)
    close(N_(ade), W.batch_ade(rec, gt), tol=2e-6)
    close(N_(fde), W.batch_fde(rec, gt), tol=2e-6)


@pytest.mark.parametrize("n,s,mode", [(1003, 20, 2), (1003, 20, 1), (1003, 20, 0), (7, 20, 2), (500, 12, 2), (333, 64, 1), (50, 5, 2)])
def test_fused_metrics_with_the_projections_pose_record(ops, oracle, dev, n, s, mode):
    """pose (5,N) -- the optional sixth output of the projection: origin, rotation x scale, 1 / scale with the moving / static
    decision in its sign -- against the oracle's normaliser, and the matrix-core metrics kernel fed from it (no nrm, no obs)
    against the oracle's reconstruction + compute_batch_ade / fde; a shape that kernel does not take (S = 5) falls back to nrm."""
    from oracle import wrapper_ref as W
    from eigentrajectory_amd.synth import synthetic_trajectories_np
    rng = np.random.default_rng(n + s + mode)
    obs, gt = synthetic_trajectories_np(n, seed=12)
    obs[::7, -3] = obs[::7, -1] + 1e-3 * rng.standard_normal((len(obs[::7]), 2)).astype(np.float32)  # (rows under static_dist)
    uo = rng.standard_normal((16, 6)).astype(np.float32) * 0.3
    um, us_ = (rng.standard_normal((24, 6)).astype(np.float32) * 0.3 for _ in range(2))
    a_m, a_s = (rng.standard_normal((6, s)).astype(np.float32) for _ in range(2))
    c = rng.standard_normal((6, n, s)).astype(np.float32)
    _, _, nrm, flag, pose = ops.norm_project(T(obs, dev), None, T(uo, dev), None, T(uo, dev), None, mode, 0.3, want_pose=True)
    ox, oy, dx, dy = N_(nrm).astype(np.float64)
    r = np.hypot(dx, dy)
    mv = N_(flag).astype(bool)
    sca = np.where(mv, 2.0 / r, 1.0)
    want = np.stack([ox, oy, dx / r * sca, dy / r * sca, np.where(mv, -1.0, 1.0) / sca])
    np.testing.assert_allclose(N_(pose), want, rtol=3e-6, atol=1e-30)
    assert np.array_equal(np.signbit(N_(pose)[4]), mv)
    kw = dict(pose=pose) if s >= 12 else dict(pose=pose, nrm=nrm)
    ade, fde = ops.anchor_reconstruct_metrics(T(c, dev), T(gt, dev), T(a_m, dev), T(a_s, dev), T(um, dev), T(us_, dev), mode,
                                              0.3, **kw)
    rec = oracle.anchor_reconstruct(c, obs, a_m, a_s, um, us_, mode, 0.3)
    close(N_(ade), W.batch_ade(rec, gt), tol=2e-6)
    close(N_(fde), W.batch_fde(rec, gt), tol=2e-6)
    if s < 12:  # the pose alone is not enough for a shape the matrix-core kernel does not take
        with pytest.raises(Exception):
            ops.anchor_reconstruct_metrics(T(c, dev), T(gt, dev), T(a_m, dev), T(a_s, dev), T(um, dev), T(us_, dev), mode, 0.3, pose=pose)


@pytest.mark.parametrize("what", ["coefficients", "U", "nan"])
def test_fused_metrics_values_beyond_f16_take_the_fp32_instructions(ops, oracle, dev, what):
    """|coefficient + anchor| >= 256 or |U| >= 32 would overflow the scaled f16 operands: those tiles (or the whole launch)
    run the fp32 matrix instructions; a NaN coefficient makes its trajectory's metrics NaN (torch.min propagates it) and
    leaves its neighbours alone."""
    from oracle import wrapper_ref as W
    from eigentrajectory_amd.synth import synthetic_trajectories_np
    n, s = 400, 20
    rng = np.random.default_rng(3)
    obs, gt = synthetic_trajectories_np(n, seed=10)
    um, us_ = (rng.standard_normal((24, 6)).astype(np.float32) * 0.3 for _ in range(2))
    a_m, a_s = (rng.standard_normal((6, s)).astype(np.float32) for _ in range(2))
    c = rng.standard_normal((6, n, s)).astype(np.float32)
    if what == "coefficients":
        c[:, 100:140] *= 3000.0  # a stretch of passes beyond the f16 range, the rest inside
    elif what == "U":
        um *= 200.0
    else:
        c[2, 57, 11] = np.nan
    ade, fde = ops.anchor_reconstruct_metrics(T(c, dev), T(gt, dev), T(a_m, dev), T(a_s, dev), T(um, dev), T(us_, dev), 2,
                                              0.3, obs=T(obs, dev))
    rec = oracle.anchor_reconstruct(c, obs, a_m, a_s, um, us_, 2, 0.3)
    want_a, want_f = W.batch_ade(rec, gt), W.batch_fde(rec, gt)
    if what == "nan":
        assert np.isnan(N_(ade)[57]) and np.isnan(N_(fde)[57]) and np.isnan(want_a[57])
        keep = np.arange(n) != 57
        close(N_(ade)[keep], want_a[keep], tol=2e-6)
        close(N_(fde)[keep], want_f[keep], tol=2e-6)
    else:
        np.testing.assert_allclose(N_(ade), want_a, rtol=3e-6, atol=2e-5)
        np.testing.assert_allclose(N_(fde), want_f, rtol=3e-6, atol=2e-5)


# ------------------------------------------------------ full-size properties (N = 1e6 and 1e7)
@pytest.mark.parametrize("n", [1_000_000, 10_000_000])
def test_full_size_properties(ops, dev, n):
    """Size-independent checks at BASELINE.json's sizes (configs 2 and 4: N = 1e6, 1e7): P(R(P(x))) = P(x); sharding
    is additive (uneven shards); Lloyd never increases the inertia; labels are the arg-max of the similarity."""
    from eigentrajectory_amd.synth import synthetic_trajectories_torch
    obs, pred = synthetic_trajectories_torch(n, dev, seed=0)
    sd = 0.3
    us = {}
    for which in (1, 0):
        g_obs, g_pred, cnt = ops.fit_gram(obs, pred, ops.MODE_SPLIT, sd, which)
        us[which] = (ops.eigh_topk(g_obs, 6)[0], ops.eigh_topk(g_pred, 6)[0])
        # additivity of the Gram over two shards (what the RCCL all-reduce relies on)
        h = n // 3
        ga, _, ca = ops.fit_gram(obs[:h], pred[:h], ops.MODE_SPLIT, sd, which)
        gb, _, cb = ops.fit_gram(obs[h:], pred[h:], ops.MODE_SPLIT, sd, which)
        assert int(ca.item() + cb.item()) == int(cnt.item())
        assert torch.allclose(ga + gb, g_obs, rtol=0, atol=1e-12 * float(g_obs.abs().max()))
        uu = us[which][1].double()
        assert torch.allclose(uu.T @ uu, torch.eye(6, device=dev, dtype=torch.float64), atol=1e-6)
    c_obs, c_pred, nrm, flag = ops.norm_project(obs, pred, us[1][0], us[1][1], us[0][0], us[0][1], ops.MODE_SPLIT, sd)
    assert torch.isfinite(c_pred).all()
    rec = ops.anchor_reconstruct(c_pred.unsqueeze(-1).contiguous(), None, None, us[1][1], us[0][1], ops.MODE_SPLIT, sd,
                                 nrm=nrm)[0]
    _, c_again, _, _ = ops.norm_project(obs, rec.contiguous(), us[1][0], us[1][1], us[0][0], us[0][1], ops.MODE_SPLIT, sd)
    scale = float(c_pred.abs().max())
    assert float((c_again - c_pred).abs().max()) < 2e-5 * scale  # projector idempotence: U^T U = I
    err = (rec - pred).norm(dim=-1).mean()
    assert float(err) < 0.2  # k=6 keeps the low-rank reconstruction error small (metres)
    x = c_pred.contiguous()
    c0 = ops.kmeans_init_farthest(x, 20, 12345)
    res = fit_and_check_traceless(ops, x, c0, 25, 1e-4)  # both forms of the one-launch-per-iteration kernel
    tr = res["trace"].cpu().numpy()
    assert (np.diff(tr[:, 1]) <= 1e-6 * tr[0, 1]).all(), "Lloyd iterations must not increase the inertia"
    counts = torch.bincount(res["labels"], minlength=20)
    assert int(counts.sum()) == n and int(counts.min()) > 0
    # the returned labels are the assignment against the centroids of the previous iteration; one more
    # fit iteration from the final centroids must reproduce predict()
    lb, ms = ops.kmeans_predict(x, res["centroids"])
    res2 = ops.kmeans_fit(x, res["centroids"], 1, 1e-4)
    assert torch.equal(res2["labels"], lb)
    assert abs(res2["inertia"] - float((-ms.double()).mean())) < 1e-5 * abs(res2["inertia"])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("n,iters", [(1_000_000, 12), (10_000_000, 5)])
def test_headline_sizes_vs_oracle(ops, oracle, dev, n, iters):
    """BASELINE.json's own sizes (configs 2 and 4) against the oracle itself, on bench.py's data and with the library's
    default options: the Gram over all rows; projection and reconstruction (S = 1) on every 64th row and the last 1000
    (rows are independent); the 20 farthest-first picks; and the exact-sum fit as the bench runs it -- trace-less, i.e.
    on the packed f16 copy with the certification of csrc/et_kmeans_packed.hip -- labels, centroids, iteration count bit
    for bit against the scalar restatement of kmeans.py:143-259 after `iters` iterations."""
    import ctypes as C
    from eigentrajectory_amd import _lib as L
    from eigentrajectory_amd.synth import synthetic_trajectories_torch
    obs, pred = synthetic_trajectories_torch(n, dev, seed=0, min_disp=1e-3)
    mode = ops.MODE_MOVING
    g_obs, g_pred, cnt = ops.fit_gram(obs, pred, mode, 0.0, 1)
    obs_np, pred_np = N_(obs), N_(pred)
    r_obs, r_pred, r_cnt = oracle.fit_gram(obs_np, pred_np, 1, 0.0, 1)
    assert int(cnt.item()) == r_cnt == n
    # (the fp32 normalised rows differ from the oracle's in the last ulp -- sincosf / atan2f of the device library vs
    # glibc --; over 1e6 .. 1e7 rows those differences average out to ~1e-9 of the largest entry: bar 2e-8.  Exact
    # summation itself: test_fit_gram_summation_exact)
    for g, r in ((g_obs, r_obs), (g_pred, r_pred)):
        close(N_(g), r, tol=2e-8)
    (U_obs, _), (U_pred, _) = ops.eigh_topk_batch([g_obs, g_pred], 6)
    for U, g in ((U_obs, g_obs), (U_pred, g_pred)):
        np.testing.assert_allclose(N_(U), oracle.eigh_topk(N_(g), 6)[0], atol=1e-6)
    c_obs, c_pred, nrm, _ = ops.norm_project(obs, pred, U_obs, U_pred, None, None, mode, want_flag=False)
    rec = ops.anchor_reconstruct(c_pred.view(6, n, 1), None, None, U_pred, None, mode, nrm=nrm)
    rows = np.unique(np.concatenate([np.arange(0, n, 64), np.arange(n - 1000, n)]))
    ro, rp, rn, _ = oracle.norm_project(obs_np[rows], pred_np[rows], N_(U_obs), N_(U_pred), None, None, 1)
    x_np = N_(c_pred)
    assert np.array_equal(N_(nrm)[:, rows], rn)
    close(N_(c_obs)[:, rows], ro)
    close(x_np[:, rows], rp)
    r_rec = oracle.anchor_reconstruct(np.ascontiguousarray(x_np[:, rows, None]), obs_np[rows], None, None, N_(U_pred), None, 1)
    close(N_(rec)[:, rows], r_rec)
    del rec, c_obs, obs, pred, obs_np, pred_np
    # k-means on the GPU's own coefficients (the same bits go to the oracle)
    first = 12345
    c0 = ops.kmeans_init_farthest(c_pred, 20, first)
    r_c0, _ = oracle.kmeans_init_farthest(x_np, 20, first)
    assert np.array_equal(N_(c0), r_c0)
    fits = L.lib().et_internal_kmeans_packed_fits
    fits.restype = C.c_longlong
    before = fits()
    res = ops.kmeans_fit(c_pred, c0, iters, 1e-4, trace=False)
    assert fits() == before + 1, "the bench's fit iterates on the packed copy: this test must too"
    ref = oracle.kmeans_fit(x_np, r_c0, iters, 1e-4)
    assert res["n_iter"] == ref["n_iter"] == iters
    assert np.array_equal(N_(res["labels"]), ref["labels"])
    assert np.array_equal(N_(res["centroids"]), ref["centroids"])
    assert np.float32(res["error"]) == np.float32(ref["error"]) and np.float32(res["inertia"]) == np.float32(ref["inertia"])


@pytest.mark.parametrize("scene", G.SCENES + ["univ-all"])
def test_sgcn_full_splits_replay_g14(dev, scene):
    """Config 3 at full extent (BASELINE.json: EigenTrajectory-SGCN inference, 20 samples, all five ETH/UCY splits, ADE/FDE
    vs the reference): EVERY test scene of eth / hotel / zara1 / zara2 and every tenth of univ's (tools/
    make_golden_sgcn_full.py: the imported reference's wrapper + sgcn bridge + its seeded SGCN) replayed through the
    PRODUCT -- wrapper (HIP projection) -> sgcn bridge contract (the network's recorded input is checked, its recorded
    output answered) -> HIP reconstruction: best-of-20 ADE / FDE per pedestrian within 1e-5 m of the reference's, the
    split-level means (utils/trainer.py:173-195) within 1e-5, for the fused metrics epilogue AND for the materialised
    trajectories of the test loop's own call `model(obs)`."""
    from eigentrajectory_amd import EigenTrajectory
    from eigentrajectory_amd.bridges import get_hook_func
    from eigentrajectory_amd.utils import default_hyper_params
    from .test_bridges import ReplaySGCN
    # "univ-all" (G14b): ALL 947 test scenes of univ, 24 334 pedestrians -- the whole of the reference's test loop on that split
    z = G.load("g14b_sgcn_univ_all.npz" if scene == "univ-all" else "g14_sgcn_full_splits.npz")
    every_scene = scene != "univ"
    scene = "univ" if scene == "univ-all" else scene
    g2 = G.load("g2_fit_all_scenes.npz")
    obs, pred, sse = G.dataset(scene, "test")
    net = ReplaySGCN(None, None, None, 2e-5)
    model = EigenTrajectory(net, get_hook_func("sgcn"), default_hyper_params(static_dist=float(z[f"{scene}.static_dist"])))
    sd = model.state_dict()
    for key in list(sd):
        if key.startswith("ET_"):
            sd[key] = torch.from_numpy(g2[f"{scene}.{key}"])
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    v_all, out_all = torch.from_numpy(z[f"{scene}.v"]), torch.from_numpy(z[f"{scene}.net_out"]).to(dev)
    sizes = z[f"{scene}.scene_size"]
    assert len(sizes) == (len(sse) if every_scene else len(range(0, len(sse), 10)))  # every scene of the split (G14's univ: index % 10 == 0)
    fused, plain, at = [], [], 0
    with torch.no_grad():
        for i, n in zip(z[f"{scene}.scene_index"], sizes):
            s, e = sse[int(i)]
            n = int(n)
            assert e - s == n
            net.expect = v_all[:, at:at + n].reshape(1, -1, n, 1)
            net.eye_shapes = np.asarray([[1, n, n], [n, 1, 1]])
            net.answer = out_all[:, at:at + n].contiguous()
            o, p = T(obs[s:e], dev), T(pred[s:e], dev)
            ade, fde = model.evaluate(o, p)
            fused.append(torch.stack([ade, fde]))
            rec = model(o)["recon_traj"]  # (S, n, 12, 2): what the reference's test loop evaluates (utils/trainer.py:183-186)
            dist = (rec - p[None]).norm(p=2, dim=-1)
            plain.append(torch.stack([dist.mean(dim=-1).min(dim=0)[0], dist[..., -1].min(dim=0)[0]]))
            at += n
    assert at == v_all.shape[1] == len(z[f"{scene}.ade"])
    ref = np.stack([z[f"{scene}.ade"], z[f"{scene}.fde"]])
    for got in (N_(torch.cat(fused, dim=1)), N_(torch.cat(plain, dim=1))):
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-5)
        np.testing.assert_allclose(got.mean(axis=1, dtype=np.float64), z[f"{scene}.ade_fde_mean"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("extent", ["tenth", "all"])
def test_agentformer_tenth_of_univ_replay_g15(dev, extent):
    """Config 5's data path (BASELINE.json: EigenTrajectory-AgentFormer, univ) at G14's extent through the PRODUCT: every tenth
    test scene of univ (95 scenes, 2 471 pedestrians; tools/make_golden_agentformer_full.py: the imported reference's wrapper
    + agentformer bridge + its seeded AgentFormerLight) -- wrapper (HIP projection) -> agentformer bridge contract (the
    predictor's recorded input is checked, its recorded output answered) -> HIP reconstruction: best-of-20 ADE / FDE per
    pedestrian and their means within 1e-5 of the reference's, for the fused metrics epilogue and for the materialised
    trajectories of the test loop's own call `model(obs)`.  extent = "all" (G15b): ALL 947 test scenes of univ, 24 334
    pedestrians -- the whole of the reference's test loop (utils/trainer.py:173-195) on config 5's split."""
    from eigentrajectory_amd import EigenTrajectory
    from eigentrajectory_amd.bridges import get_hook_func
    from eigentrajectory_amd.utils import default_hyper_params
    from .test_bridges import ReplayAgentFormer
    z = G.load("g15b_agentformer_univ_all.npz" if extent == "all" else "g15_agentformer_univ_tenth.npz")
    g2 = G.load("g2_fit_all_scenes.npz")
    obs, pred, sse = G.dataset("univ", "test")
    net = ReplayAgentFormer(None, None, 2e-5)
    model = EigenTrajectory(net, get_hook_func("agentformer"), default_hyper_params(static_dist=float(z["static_dist"])))
    sd = model.state_dict()
    for key in list(sd):
        if key.startswith("ET_"):
            sd[key] = torch.from_numpy(g2[f"univ.{key}"])
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    pre_all, dec_all = torch.from_numpy(z["pre_motion"]), torch.from_numpy(z["dec_motion"]).to(dev)
    assert list(z["scene_index"]) == list(range(0, len(sse), 1 if extent == "all" else 10))
    fused, plain, at = [], [], 0
    with torch.no_grad():
        for i, n in zip(z["scene_index"], z["scene_size"]):
            s, e = sse[int(i)]
            n = int(n)
            assert e - s == n
            net.expect, net.answer = pre_all[:, at:at + n], dec_all[at:at + n]
            o, p = T(obs[s:e], dev), T(pred[s:e], dev)
            ade, fde = model.evaluate(o, p)
            fused.append(torch.stack([ade, fde]))
            rec = model(o)["recon_traj"]  # (S, n, 12, 2): what the reference's test loop evaluates (utils/trainer.py:183-186)
            dist = (rec - p[None]).norm(p=2, dim=-1)
            plain.append(torch.stack([dist.mean(dim=-1).min(dim=0)[0], dist[..., -1].min(dim=0)[0]]))
            at += n
    assert at == len(z["ade"]) == (24334 if extent == "all" else 2471)
    ref = np.stack([z["ade"], z["fde"]])
    for got in (N_(torch.cat(fused, dim=1)), N_(torch.cat(plain, dim=1))):
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-5)
        np.testing.assert_allclose(got.mean(axis=1, dtype=np.float64), z["ade_fde_mean"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("n,t_obs,t_pred,mode", [(70001, 8, 12, 2), (257, 8, 12, 1), (1, 8, 12, 0), (5000, 5, 7, 1), (3000, 8, 12, 3)])
def test_fit_descriptor_one_call_equals_gram_plus_eigh(ops, dev, n, t_obs, t_pred, mode):
    """et_fit_descriptor (descriptor.py:116-142 in one call: the Gram kernel, its partial reduction and ONE launch that assembles and
    solves both eigenproblems) returns the bits of et_fit_gram followed by et_eigh_topk_batch -- G, count, U, sigma --, for the
    (8, 12) fast path and for any other shape (which runs the two calls one after the other)."""
    rng = np.random.default_rng(n)
    obs = (rng.standard_normal((n, t_obs, 2)).cumsum(1) * 0.3).astype(np.float32)
    pred = (obs[:, -1:, :] + rng.standard_normal((n, t_pred, 2)).cumsum(1) * 0.3).astype(np.float32)
    k = min(6, 2 * t_obs)
    for which in (1, 0):
        g_obs, g_pred, cnt = ops.fit_gram(T(obs, dev), T(pred, dev), mode, 0.3, which)
        (U_obs, s_obs), (U_pred, s_pred) = ops.eigh_topk_batch([g_obs, g_pred], k)
        got = ops.fit_descriptor(T(obs, dev), T(pred, dev), k, mode, 0.3, which, want_gram=True)
        for a, b in zip(got, (U_obs, U_pred, s_obs, s_pred, cnt, g_obs, g_pred)):
            assert np.array_equal(N_(a), N_(b), equal_nan=True)
        lean = ops.fit_descriptor(T(obs, dev), T(pred, dev), k, mode, 0.3, which)  # (no G outputs requested)
        assert torch.equal(lean[0], U_obs) and torch.equal(lean[1], U_pred) and torch.equal(lean[4], cnt)
