"""Predictor plug-in contracts ("bridges") for the ten baselines the reference supports.

The reference resolves, per baseline, three hooks (`trainval.py:24-28`, `baseline/<name>/bridge.py`):

    input_data    = model_forward_pre_hook(C_obs (k,N), obs_ori (2,N), addl_info)
    output_data   = model_forward(input_data, baseline_model)
    C_pred_refine = model_forward_post_hook(output_data, addl_info)        # must be (k,N,S)

and the wrapper calls them between projection and reconstruction (`EigenTrajectory/model.py:93-95`).
The predictors themselves are third-party networks and out of scope here;
what this module keeps is the *contract*, so that any of them plugs into
:class:`eigentrajectory_amd.EigenTrajectory` unchanged::

    hooks = get_hook_func("sgcn")           # DotDict with the three callables
    model = EigenTrajectory(predictor, hooks, hyper_params)

The ten bridges differ only in how the (k+2, N) coefficient block is laid out for the network and in
which axes the network's output uses, so they are expressed here as one table of small layout
functions instead of ten files.  Host-side tensor glue: device agnostic, no kernels.
"""
from __future__ import annotations

from collections import defaultdict

import torch

from .utils import DotDict


# ----------------------------------------------------------------------------- input layouts
def _stack(C_obs, obs_ori):
    """(k,N) coefficients + (2,N) scene-centred last position -> (k+2, N)  (e.g. sgcn/bridge.py:5-6)"""
    return C_obs if obs_ori is None else torch.cat([C_obs, obs_ori], dim=0)


def _tvn(x):
    """(T,N) -> (1, T, N, 1), detached: 'batch, time, vertex, channel' (sgcn/bridge.py:7)"""
    return x[None, :, :, None].detach()


def _ctn(x):
    """(T,N) -> (1, 1, T, N): 'batch, channel, time, vertex' (stgcnn/bridge.py:28-29)"""
    return _tvn(x).permute(0, 3, 1, 2)


def _pairwise_norm(v):
    """v (1,C,T,N) -> (1,T,N,N) Euclidean distance between vertices per time step"""
    t = v.permute(0, 2, 3, 1)                      # (1,T,N,C)
    return (t.unsqueeze(-2) - t.unsqueeze(-3)).norm(p=2, dim=-1)


def _stgcnn_adjacency(v):
    """Normalised graph Laplacian of the inverse-distance kernel (stgcnn/bridge.py:4-21)."""
    n = v.size(-1)
    dist = _pairwise_norm(v)
    a_inv = 1.0 / dist
    a_inv[dist == 0] = 0
    eye = torch.eye(n, device=v.device)
    a_hat = a_inv + eye
    deg = a_hat.sum(dim=-1, keepdim=True).pow(-0.5)
    deg[torch.isinf(deg)] = 0
    d = eye * deg
    return eye - d @ a_hat @ d


def _dmrgcn_adjacency(v):
    """Stack of displacement- and distance- adjacency (dmrgcn/bridge.py:4-19)."""
    rel = torch.zeros_like(v)
    rel[..., 1:, :] = v[..., 1:, :] - v[..., :-1, :]
    return torch.stack([_pairwise_norm(rel), _pairwise_norm(v)], dim=1)


def _pre_sgcn(C_obs, obs_ori, addl_info=None):
    v = _tvn(_stack(C_obs, obs_ori))
    t, n = v.size(1), v.size(2)
    eyes = [torch.eye(n, device=v.device).repeat(v.size(3), 1, 1), torch.eye(v.size(3), device=v.device).repeat(n, 1, 1)]
    del t
    return v, eyes


def _pre_stgcnn(C_obs, obs_ori, addl_info=None):
    v = _ctn(_stack(C_obs, obs_ori))
    return v, _stgcnn_adjacency(v).squeeze(dim=0).detach()


def _pre_dmrgcn(C_obs, obs_ori, addl_info=None):
    v = _ctn(_stack(C_obs, obs_ori))
    return v, _dmrgcn_adjacency(v).detach()


def _pre_gpgraph_stgcnn(C_obs, obs_ori, addl_info=None):
    v = _ctn(_stack(C_obs, obs_ori))
    return v, v  # the coefficients themselves stand in for the velocity input (gpgraphstgcnn/bridge.py:12)


def _pre_gpgraph_sgcn(C_obs, obs_ori, addl_info=None):
    v = _ctn(_stack(C_obs, obs_ori))
    pos = torch.arange(1, v.size(2) + 1, device=v.device, dtype=v.dtype).unsqueeze(-1).expand_as(v)
    return v, torch.cat([pos, v], dim=1)  # prepend the 1..T position channel (gpgraphsgcn/bridge.py:13-15)


def _pre_graphtern(C_obs, obs_ori, addl_info=None):
    a = _tvn(_stack(C_obs, obs_ori))
    r = torch.zeros_like(a)
    r[:, 1:] = a[:, 1:] - a[:, :-1]
    return (torch.stack([a, r], dim=1),)


def _pre_implicit(C_obs, obs_ori, addl_info=None):
    return (_ctn(_stack(C_obs, obs_ori)),)


def _pre_pecnet(C_obs, obs_ori, addl_info):
    return C_obs.T, obs_ori.T, addl_info["scene_mask"], obs_ori.T


def _pre_lbebm(C_obs, obs_ori, addl_info=None):
    return C_obs.T, obs_ori.T


def _pre_agentformer(C_obs, obs_ori, addl_info=None):
    data = defaultdict(lambda: None)
    data["pre_motion"] = _stack(C_obs, obs_ori).unsqueeze(dim=-1).contiguous()
    return data


# ---------------------------------------------------------------------------------- forwards
def _call(input_data, baseline_model):
    return baseline_model(*input_data)


def _call_predict(input_data, baseline_model):
    return baseline_model.predict(*input_data)


def _call_agentformer(input_data, baseline_model):
    baseline_model.set_data(input_data)
    baseline_model()
    return baseline_model.data


# ---------------------------------------------------------------------------- output layouts
def _post_identity(out, addl_info=None):
    return out


def _post_squeeze(out, addl_info=None):
    return out.squeeze(dim=0)


def _post_sktn(out, addl_info=None):
    """(1, S, k, N) -> (k, N, S)"""
    return out.permute(0, 2, 3, 1).squeeze(dim=0)


def _post_first_sktn(out, addl_info=None):
    return _post_sktn(out[0])


def _post_flat_samples(out, addl_info):
    """(N, k*S) -> (k, N, S)  (pecnet/bridge.py:13-17)"""
    n, ks = out.shape
    s = addl_info["num_samples"]
    return out.view(n, ks // s, s).permute(1, 0, 2)


def _post_agentformer(out, addl_info=None):
    return out["_dec_motion"].permute(1, 0, 2)


#: baseline name -> (pre hook, forward, post hook); the names are the reference's package names
BRIDGES = {
    "stgcnn": (_pre_stgcnn, _call, _post_sktn),
    "sgcn": (_pre_sgcn, _call, _post_identity),
    "pecnet": (_pre_pecnet, _call_predict, _post_flat_samples),
    "agentformer": (_pre_agentformer, _call_agentformer, _post_agentformer),
    "lbebm": (_pre_lbebm, _call_predict, _post_flat_samples),
    "dmrgcn": (_pre_dmrgcn, _call, _post_first_sktn),
    "gpgraphsgcn": (_pre_gpgraph_sgcn, _call, _post_first_sktn),
    "gpgraphstgcnn": (_pre_gpgraph_stgcnn, _call, _post_first_sktn),
    "graphtern": (_pre_graphtern, _call, _post_squeeze),
    "implicit": (_pre_implicit, _call, _post_sktn),
}


def get_hook_func(baseline: str) -> DotDict:
    """The DotDict the reference builds at trainval.py:25-27 for ``baseline``."""
    try:
        pre, fwd, post = BRIDGES[baseline]
    except KeyError:
        raise ValueError(f"unknown baseline {baseline!r}; known: {sorted(BRIDGES)}") from None
    return DotDict({"model_forward_pre_hook": pre, "model_forward": fwd, "model_forward_post_hook": post})
