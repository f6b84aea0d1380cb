"""Loaders for the committed golden fixtures (tests/golden/, made by tools/make_golden.py)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCENES = ["eth", "hotel", "univ", "zara1", "zara2"]


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def manifest():
    with open(os.path.join(GOLDEN, "MANIFEST.json")) as f:
        return json.load(f)


def dataset(scene, phase):
    """-> obs (N,8,2) f32, pred (N,12,2) f32, seq_start_end (M,2) int32 (utils/dataloader.py windows)."""
    z = np.load(os.path.join(GOLDEN, "data", f"{scene}_{phase}.npz"))
    full = (z["q"].astype(np.float64) / 1e4).astype(np.float32)
    return np.ascontiguousarray(full[:, :8]), np.ascontiguousarray(full[:, 8:]), z["seq_start_end"]


def eth_fit_input():
    """ETH train+val with the y-flip augmentation (utils/trainer.py:51-53, utils/utils.py:79-81)."""
    o1, p1, _ = dataset("eth", "train")
    o2, p2, _ = dataset("eth", "val")
    obs, pred = np.concatenate([o1, o2]), np.concatenate([p1, p2])
    flip = np.array([[[1.0, -1.0]]], dtype=np.float32)
    return np.concatenate([obs, obs * flip]), np.concatenate([pred, pred * flip])


def static_dist(scene):
    return float(manifest()["static_dist"][scene])


def sign_align(U, U_ref):
    """Flip columns of U to the sign of U_ref (LAPACK's SVD signs are arbitrary)."""
    s = np.sign((U * U_ref).sum(axis=0))
    s[s == 0] = 1
    return U * s
