#!/usr/bin/env python3
"""Golden vectors for the ten baseline bridges (baseline/*/bridge.py of the reference): seeded
(C_obs, obs_ori) -> pre-hook outputs, and seeded predictor outputs -> post-hook outputs.
Run in the build container only:  python tools/make_golden_bridges.py"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g10_bridges.npz")
NAMES = ["stgcnn", "sgcn", "pecnet", "agentformer", "lbebm", "dmrgcn", "gpgraphsgcn", "gpgraphstgcnn", "graphtern", "implicit"]


def load_bridge(name):  # import the bridge file alone (the package __init__ pulls in every predictor)
    spec = importlib.util.spec_from_file_location(f"ref_bridge_{name}", os.path.join(REF, "baseline", name, "bridge.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def flat(prefix, obj, out):
    if isinstance(obj, torch.Tensor):
        out[prefix] = obj.numpy()
    elif isinstance(obj, (tuple, list)):
        out[prefix + ".len"] = np.int64(len(obj))
        for i, o in enumerate(obj):
            flat(f"{prefix}.{i}", o, out)
    elif isinstance(obj, dict):
        for k in ("pre_motion",):
            flat(f"{prefix}.{k}", obj[k], out)
        out[prefix + ".missing_is_none"] = np.bool_(obj["anything_else"] is None)
    else:
        raise TypeError(type(obj))


def main():
    g = torch.Generator().manual_seed(5)
    k, n, s = 6, 7, 20
    c_obs = torch.randn(k, n, generator=g)
    obs_ori = torch.randn(2, n, generator=g)
    scene_mask = torch.zeros(n, n, dtype=torch.bool)
    scene_mask[:3, :3] = True
    scene_mask[3:, 3:] = True
    addl = {"scene_mask": scene_mask, "num_samples": s}
    out = dict(C_obs=c_obs.numpy(), obs_ori=obs_ori.numpy(), scene_mask=scene_mask.numpy())
    # what each predictor family returns (shapes from the trainers' constructor arguments, utils/trainer.py:267-562)
    raw = {
        "stgcnn": torch.randn(1, s, k, n, generator=g), "implicit": torch.randn(1, s, k, n, generator=g),
        "sgcn": torch.randn(k, n, s, generator=g), "graphtern": torch.randn(1, k, n, s, generator=g),
        "pecnet": torch.randn(n, k * s, generator=g), "lbebm": torch.randn(n, k * s, generator=g),
        "agentformer": torch.randn(n, k, s, generator=g),
        "dmrgcn": torch.randn(1, s, k, n, generator=g), "gpgraphsgcn": torch.randn(1, s, k, n, generator=g),
        "gpgraphstgcnn": torch.randn(1, s, k, n, generator=g),
    }
    for name in NAMES:
        b = load_bridge(name)
        flat(f"{name}.pre", b.model_forward_pre_hook(c_obs, obs_ori, addl), out)
        r = raw[name]
        out[f"{name}.raw"] = r.numpy()
        if name == "agentformer":
            post_in = {"_dec_motion": r}
        elif name in ("dmrgcn", "gpgraphsgcn", "gpgraphstgcnn"):
            post_in = (r, None)
        else:
            post_in = r
        out[f"{name}.post"] = b.model_forward_post_hook(post_in, addl).contiguous().numpy()
        assert out[f"{name}.post"].shape == (k, n, s), (name, out[f"{name}.post"].shape)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
