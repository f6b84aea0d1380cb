"""The PyTorch-CPU restatement that bench.py times as `cpu_baseline` (oracle/torch_cpu_ref.py) against the golden
vectors captured from the reference (tests/golden/, G1 / G2 / G4 / G5 / G7): it must compute what the reference
computes, otherwise its timing is not a baseline for anything."""
import numpy as np
import pytest
import torch

from . import _golden as G

FP = dict(rtol=1e-5, atol=1e-5)


@pytest.fixture(scope="module")
def R():
    from oracle import torch_cpu_ref
    torch.set_num_threads(1)
    return torch_cpu_ref


@pytest.mark.parametrize("sca", [True, False])
def test_trajnorm_g1(R, sca):
    g1 = G.load("g1_trajnorm_eth_test.npz")
    obs, pred, _ = G.dataset("eth", "test")
    o, p = torch.from_numpy(obs), torch.from_numpy(pred)
    t = "sca1" if sca else "sca0"
    ori, rot, s = R.norm_params(o, sca)
    assert np.array_equal(ori.numpy(), g1[t + "_ori"]) and np.array_equal(rot.numpy(), g1[t + "_rot"])
    fin = np.isfinite(g1[t + "_pred_norm"]).all(axis=(1, 2))
    pn = R.normalize(p, ori, rot, s)
    assert np.array_equal(pn.numpy()[fin], g1[t + "_pred_norm"][fin])  # the same ATen ops: the same bits
    assert np.array_equal(R.denormalize(pn, ori, rot, s).numpy()[fin], g1[t + "_pred_roundtrip"][fin])


@pytest.mark.parametrize("tag,sca", [("m", True), ("s", False)])
def test_fit_project_reconstruct_g2_g4_g5(R, tag, sca):
    g2, z = G.load("g2_fit_all_scenes.npz"), G.load("g45_project_reconstruct_eth_test.npz")
    # fit: SVD of the ETH fit rows routed to this descriptor (model.py:46-52)
    obs, pred = (torch.from_numpy(a) for a in G.eth_fit_input())
    moving = (obs[:, -1] - obs[:, -3]).div(2).norm(p=2, dim=-1) > G.static_dist("eth")
    sel = moving if sca else ~moving
    ori, rot, s = R.norm_params(obs[sel], sca)
    U_pred = R.truncated_svd(R.normalize(pred[sel], ori, rot, s), 6)[0].numpy()
    U_ref = g2[f"eth.ET_{tag}_descriptor.U_pred_trunc"]
    np.testing.assert_allclose(G.sign_align(U_pred, U_ref), U_ref, atol=2e-6)
    # projection / reconstruction on ETH test with the reference's U
    o, p, _ = G.dataset("eth", "test")
    rows = z[f"{tag}.rows"]
    o, p = torch.from_numpy(o[rows]), torch.from_numpy(p[rows])
    Uo, Up = (torch.from_numpy(g2[f"eth.ET_{tag}_descriptor.{k}"]) for k in ("U_obs_trunc", "U_pred_trunc"))
    A = torch.from_numpy(g2[f"eth.ET_{tag}_anchor.C_anchor"])
    ori, rot, s = R.norm_params(o, sca)
    np.testing.assert_allclose(R.to_et_space(R.normalize(o, ori, rot, s), Uo).numpy(), z[f"{tag}.C_obs"], **FP)
    np.testing.assert_allclose(R.to_et_space(R.normalize(p, ori, rot, s), Up).numpy(), z[f"{tag}.C_pred"], **FP)
    C = torch.from_numpy(z[f"{tag}.C_refine"]) + A[:, None, :]  # anchor.py:87
    np.testing.assert_allclose(R.reconstruction(C, Up, ori, rot, s).numpy(), z[f"{tag}.recon"], rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("tag", ["gauss1000", "blobs10000", "ethm"])
def test_batchkmeans_g7(R, tag):
    from eigentrajectory_amd.synth import gaussian_points_np
    g7 = G.load("g7_batchkmeans.npz")
    if tag == "ethm":
        x = g7["ethm.x"]
    else:
        x = gaussian_points_np(6, int(tag[5:]), seed=11, n_blobs=int(g7[f"{tag}.blobs"]))
    x = torch.from_numpy(np.ascontiguousarray(x))
    c0 = R.farthest_first(x, 20, int(g7[f"{tag}.first_index"]))
    assert np.array_equal(c0.numpy(), g7[f"{tag}.c0"])
    res = R.lloyd(x, c0, 100, 1e-4)
    trace = g7[f"{tag}.trace"]
    assert res["n_iter"] == len(trace)
    np.testing.assert_allclose(np.asarray(res["trace"])[:, 1], trace[:, 1], rtol=1e-5)
    assert (res["labels"].numpy() == g7[f"{tag}.labels"]).mean() > 0.999
    np.testing.assert_allclose(res["centroids"].numpy(), g7[f"{tag}.centroids"], rtol=1e-4, atol=1e-4)


def test_hot_path_runs_and_reports_stages(R):
    from eigentrajectory_amd.synth import synthetic_trajectories_np
    obs, pred = (torch.from_numpy(a) for a in synthetic_trajectories_np(4000, seed=0, min_disp=1e-3))
    stages = {}
    out = R.hot_path(obs, pred, first_index=2000, max_iter=10, stages=stages)
    assert out["recon"].shape == (1, 4000, 12, 2) and out["kmeans"]["labels"].shape == (4000,)
    assert float((out["recon"][0] - pred).norm(dim=-1).mean()) < 0.2
    assert set(stages) >= {"fit", "project", "reconstruct", "kmeans_init", "kmeans_lloyd", "total"}
