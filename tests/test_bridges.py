"""The ten baseline-bridge contracts against golden vectors captured from the reference's
baseline/*/bridge.py (tools/make_golden_bridges.py).  Host-side tensor glue: runs on CPU."""
import numpy as np
import pytest
import torch

from . import _golden as G

NAMES = ["stgcnn", "sgcn", "pecnet", "agentformer", "lbebm", "dmrgcn", "gpgraphsgcn", "gpgraphstgcnn", "graphtern",
         "implicit"]


def collect(prefix, obj, out):
    if isinstance(obj, torch.Tensor):
        out[prefix] = obj.numpy()
    elif isinstance(obj, (tuple, list)):
        out[prefix + ".len"] = np.int64(len(obj))
        for i, o in enumerate(obj):
            collect(f"{prefix}.{i}", o, out)
    else:  # agentformer's defaultdict
        out[prefix + ".pre_motion"] = obj["pre_motion"].numpy()
        out[prefix + ".missing_is_none"] = np.bool_(obj["anything_else"] is None)


@pytest.mark.parametrize("name", NAMES)
def test_bridge_matches_reference(name):
    from eigentrajectory_amd.bridges import get_hook_func
    z = G.load("g10_bridges.npz")
    hooks = get_hook_func(name)
    addl = {"scene_mask": torch.from_numpy(z["scene_mask"]), "num_samples": 20}
    got = {}
    collect(f"{name}.pre", hooks.model_forward_pre_hook(torch.from_numpy(z["C_obs"]), torch.from_numpy(z["obs_ori"]), addl),
            got)
    ref_keys = [k for k in z.files if k.startswith(f"{name}.pre")]
    assert sorted(got) == sorted(ref_keys)
    for k in ref_keys:
        np.testing.assert_allclose(got[k], z[k], rtol=1e-6, atol=1e-6, err_msg=k)
    raw = torch.from_numpy(z[f"{name}.raw"])
    post_in = {"_dec_motion": raw} if name == "agentformer" else ((raw, None) if name in ("dmrgcn", "gpgraphsgcn", "gpgraphstgcnn") else raw)
    post = hooks.model_forward_post_hook(post_in, addl)
    assert tuple(post.shape) == (6, 7, 20)
    assert np.array_equal(post.contiguous().numpy(), z[f"{name}.post"])


def test_hook_protocol_drives_a_predictor():
    from eigentrajectory_amd.bridges import BRIDGES, get_hook_func
    assert sorted(BRIDGES) == sorted(NAMES)
    with pytest.raises(ValueError):
        get_hook_func("nope")

    class Net(torch.nn.Module):  # SGCN-shaped stand-in: (1,T,N,1) + identity stacks -> (k,N,S)
        def forward(self, v, eyes):
            assert v.shape[0] == 1 and v.shape[3] == 1 and len(eyes) == 2
            return v[0, :6, :, 0, None].expand(6, v.shape[2], 20)

    hooks = get_hook_func("sgcn")
    c, o = torch.randn(6, 5), torch.randn(2, 5)
    out = hooks.model_forward_post_hook(hooks.model_forward(hooks.model_forward_pre_hook(c, o), Net()))
    assert out.shape == (6, 5, 20) and torch.equal(out[:, :, 3], c)


class ReplayAgentFormer(torch.nn.Module):
    """Stands in for AgentFormerLight (third-party predictor, not on the path): checks that the bridge hands it what
    the reference's bridge handed the real network, and answers with the real network's recorded output."""

    def __init__(self, pre_motion, dec_motion, tol):
        super().__init__()
        self.expect, self.answer, self.tol, self.data = pre_motion, dec_motion, tol, None

    def set_data(self, data):
        got = data["pre_motion"]
        assert data["anything_else"] is None and got.shape == self.expect.shape
        assert torch.allclose(got.cpu(), self.expect, rtol=0, atol=self.tol), float((got.cpu() - self.expect).abs().max())
        self._dev = got.device

    def forward(self):
        self.data = {"_dec_motion": self.answer.to(self._dev)}


def test_agentformer_end_to_end_replay_through_the_oracle_g12(oracle):
    """Config 5's data path on CPU: reference-fitted univ descriptors -> oracle projection -> THIS build's agentformer
    bridge -> recorded AgentFormerLight output -> oracle reconstruction == what the reference's wrapper produced."""
    from eigentrajectory_amd.bridges import get_hook_func
    from oracle import wrapper_ref as W
    z = G.load("g12_agentformer_univ.npz")
    g2 = G.load("g2_fit_all_scenes.npz")
    params = {k[len("univ."):]: g2[k] for k in g2.files if k.startswith("univ.ET_")}
    obs, pred, sse = G.dataset("univ", "test")
    hooks = get_hook_func("agentformer")
    for j in range(3):
        s, e = sse[int(z[f"scene{j}.index"])]
        net = ReplayAgentFormer(torch.from_numpy(z[f"scene{j}.pre_motion"]), torch.from_numpy(z[f"scene{j}.dec_motion"]), 2e-5)

        def predictor(x):  # x = cat(C_obs, obs_ori) (k+2, N), the oracle wrapper's stand-in for the pre-hook input
            xt = torch.from_numpy(x)
            data = hooks.model_forward_pre_hook(xt[:6], xt[6:], None)
            return hooks.model_forward_post_hook(hooks.model_forward(data, net), None).contiguous().numpy()
        out = W.forward(params, obs[s:e], pred[s:e], predictor, float(z["static_dist"]))
        ref = z[f"scene{j}.recon_traj"]
        np.testing.assert_allclose(out["recon_traj"], ref, rtol=0, atol=2e-5 * np.abs(ref).max())
        np.testing.assert_allclose(W.batch_ade(out["recon_traj"], pred[s:e]), z[f"scene{j}.ade"], atol=1e-5)
        np.testing.assert_allclose(W.batch_fde(out["recon_traj"], pred[s:e]), z[f"scene{j}.fde"], atol=1e-5)
        losses = [out["loss_eigentraj"], out["loss_euclidean_ade"], out["loss_euclidean_fde"]]
        np.testing.assert_allclose(losses, z[f"scene{j}.losses"], rtol=1e-5, atol=1e-5)


class ReplaySGCN(torch.nn.Module):
    """Stands in for the reference's SGCN (third-party predictor, not on the path): checks that the bridge hands it what
    the reference's bridge handed the real network -- the graph `v` (1, k+2, N, 1) and the two identity stacks,
    sgcn/bridge.py:4-12 -- and answers with the real network's recorded output (k, N, S)."""

    def __init__(self, v, eye_shapes, net_out, tol):
        super().__init__()
        self.expect, self.eye_shapes, self.answer, self.tol = v, eye_shapes, net_out, tol

    def forward(self, v, eyes):
        assert v.shape == self.expect.shape and not v.requires_grad
        assert torch.allclose(v.cpu(), self.expect, rtol=0, atol=self.tol), float((v.cpu() - self.expect).abs().max())
        assert [list(e.shape) for e in eyes] == self.eye_shapes.tolist()
        for e in eyes:
            assert torch.equal(e.cpu(), torch.eye(e.size(-1)).expand_as(e))
        return self.answer.to(v.device)


@pytest.mark.parametrize("scene", G.SCENES)
def test_sgcn_end_to_end_replay_through_the_oracle_g13(oracle, scene):
    """Config 3's data path on CPU: reference-fitted descriptors -> oracle projection -> THIS build's sgcn bridge ->
    the recorded output of the reference's SGCN -> oracle reconstruction == what the reference's wrapper + bridge +
    network produced on the same test scenes (tools/make_golden_sgcn.py), ADE / FDE within 1e-5."""
    from eigentrajectory_amd.bridges import get_hook_func
    from oracle import wrapper_ref as W
    z = G.load("g13_sgcn_all_scenes.npz")
    g2 = G.load("g2_fit_all_scenes.npz")
    params = {k[len(scene) + 1:]: g2[k] for k in g2.files if k.startswith(f"{scene}.ET_")}
    obs, pred, sse = G.dataset(scene, "test")
    hooks = get_hook_func("sgcn")
    for j in range(3):
        tag = f"{scene}.scene{j}"
        s, e = sse[int(z[f"{tag}.index"])]
        net = ReplaySGCN(torch.from_numpy(z[f"{tag}.v"]), z[f"{tag}.eye_shapes"], torch.from_numpy(z[f"{tag}.net_out"]), 2e-5)

        def predictor(x):  # x = cat(C_obs, obs_ori) (k+2, N), the oracle wrapper's stand-in for the pre-hook input
            xt = torch.from_numpy(x)
            data = hooks.model_forward_pre_hook(xt[:6], xt[6:], None)
            return hooks.model_forward_post_hook(hooks.model_forward(data, net), None).contiguous().numpy()
        out = W.forward(params, obs[s:e], pred[s:e], predictor, float(z[f"{scene}.static_dist"]))
        ref = z[f"{tag}.recon_traj"]
        np.testing.assert_allclose(out["recon_traj"], ref, rtol=0, atol=2e-5 * np.abs(ref).max())
        np.testing.assert_allclose(W.batch_ade(out["recon_traj"], pred[s:e]), z[f"{tag}.ade"], atol=1e-5)
        np.testing.assert_allclose(W.batch_fde(out["recon_traj"], pred[s:e]), z[f"{tag}.fde"], atol=1e-5)
        losses = [out["loss_eigentraj"], out["loss_euclidean_ade"], out["loss_euclidean_fde"]]
        np.testing.assert_allclose(losses, z[f"{tag}.losses"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("scene", G.SCENES + ["univ-all"])
def test_sgcn_full_splits_replay_through_the_oracle_g14(oracle, scene):
    """Config 3 at full extent on CPU (tests/golden/g14: every test scene of eth / hotel / zara1 / zara2, every tenth of
    univ's): oracle projection -> THIS build's sgcn bridge -> the recorded output of the reference's SGCN -> oracle
    reconstruction: per-pedestrian best-of-20 ADE / FDE and the split-level means within 1e-5 of the reference's."""
    from eigentrajectory_amd.bridges import get_hook_func
    from oracle import wrapper_ref as W
    # "univ-all" (G14b): ALL 947 test scenes of univ, 24 334 pedestrians
    z = G.load("g14b_sgcn_univ_all.npz" if scene == "univ-all" else "g14_sgcn_full_splits.npz")
    scene = "univ" if scene == "univ-all" else scene
    g2 = G.load("g2_fit_all_scenes.npz")
    params = {k[len(scene) + 1:]: g2[k] for k in g2.files if k.startswith(f"{scene}.ET_")}
    obs, pred, sse = G.dataset(scene, "test")
    hooks = get_hook_func("sgcn")
    v_all, out_all = torch.from_numpy(z[f"{scene}.v"]), torch.from_numpy(z[f"{scene}.net_out"])
    net = ReplaySGCN(None, None, None, 2e-5)

    def predictor(x):  # x = cat(C_obs, obs_ori) (k+2, N), the oracle wrapper's stand-in for the pre-hook input
        xt = torch.from_numpy(x)
        data = hooks.model_forward_pre_hook(xt[:6], xt[6:], None)
        return hooks.model_forward_post_hook(hooks.model_forward(data, net), None).contiguous().numpy()

    ades, fdes, at = [], [], 0
    for i, n in zip(z[f"{scene}.scene_index"], z[f"{scene}.scene_size"]):
        s, e = sse[int(i)]
        n = int(n)
        net.expect = v_all[:, at:at + n].reshape(1, -1, n, 1)
        net.eye_shapes = np.asarray([[1, n, n], [n, 1, 1]])
        net.answer = out_all[:, at:at + n].contiguous()
        out = W.forward(params, obs[s:e], None, predictor, float(z[f"{scene}.static_dist"]))
        ades.append(W.batch_ade(out["recon_traj"], pred[s:e]))
        fdes.append(W.batch_fde(out["recon_traj"], pred[s:e]))
        at += n
    ade, fde = np.concatenate(ades), np.concatenate(fdes)
    np.testing.assert_allclose(ade, z[f"{scene}.ade"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(fde, z[f"{scene}.fde"], rtol=0, atol=1e-5)
    np.testing.assert_allclose([ade.mean(dtype=np.float64), fde.mean(dtype=np.float64)], z[f"{scene}.ade_fde_mean"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("extent", ["tenth", "all"])
def test_agentformer_tenth_of_univ_replay_through_the_oracle_g15(oracle, extent):
    """Config 5's data path at G14's extent on CPU (tests/golden/g15: every tenth test scene of univ, 2 471 pedestrians;
    tools/make_golden_agentformer_full.py): oracle projection -> THIS build's agentformer bridge -> the recorded output of
    the reference's AgentFormerLight -> oracle reconstruction: per-pedestrian best-of-20 ADE / FDE and their means within
    1e-5 of the reference's."""
    from eigentrajectory_amd.bridges import get_hook_func
    from oracle import wrapper_ref as W
    # extent = "all" (G15b): ALL 947 test scenes of univ, 24 334 pedestrians
    z = G.load("g15b_agentformer_univ_all.npz" if extent == "all" else "g15_agentformer_univ_tenth.npz")
    g2 = G.load("g2_fit_all_scenes.npz")
    params = {k[len("univ."):]: g2[k] for k in g2.files if k.startswith("univ.ET_")}
    obs, pred, sse = G.dataset("univ", "test")
    hooks = get_hook_func("agentformer")
    assert int(z["pre_axis"]) == 1 and int(z["dec_axis"]) == 0 and int(z["dec_per_pedestrian"]) == 1
    pre_all, dec_all = torch.from_numpy(z["pre_motion"]), torch.from_numpy(z["dec_motion"])
    net = ReplayAgentFormer(None, None, 2e-5)

    def predictor(x):  # x = cat(C_obs, obs_ori) (k+2, N), the oracle wrapper's stand-in for the pre-hook input
        xt = torch.from_numpy(x)
        data = hooks.model_forward_pre_hook(xt[:6], xt[6:], None)
        return hooks.model_forward_post_hook(hooks.model_forward(data, net), None).contiguous().numpy()

    ades, fdes, at = [], [], 0
    for i, n in zip(z["scene_index"], z["scene_size"]):
        s, e = sse[int(i)]
        n = int(n)
        assert e - s == n
        net.expect, net.answer = pre_all[:, at:at + n], dec_all[at:at + n]
        out = W.forward(params, obs[s:e], None, predictor, float(z["static_dist"]))
        ades.append(W.batch_ade(out["recon_traj"], pred[s:e]))
        fdes.append(W.batch_fde(out["recon_traj"], pred[s:e]))
        at += n
    assert at == len(z["ade"])
    ade, fde = np.concatenate(ades), np.concatenate(fdes)
    np.testing.assert_allclose(ade, z["ade"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(fde, z["fde"], rtol=0, atol=1e-5)
    np.testing.assert_allclose([ade.mean(dtype=np.float64), fde.mean(dtype=np.float64)], z["ade_fde_mean"], rtol=0, atol=1e-5)
