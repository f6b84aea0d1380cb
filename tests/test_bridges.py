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
