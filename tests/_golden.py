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


def _windows(z):
    full = (z["q"].astype(np.float64) / 1e4).astype(np.float32)
    return np.ascontiguousarray(full[:, :8]), np.ascontiguousarray(full[:, 8:]), z["seq_start_end"]


def split_files(scene, phase):
    """File names of a train/val split in the order the reference's os.listdir gave them when the fixtures
    were captured (tools/make_golden_fitsets.py)."""
    with open(os.path.join(GOLDEN, "data", "splits.json")) as f:
        return json.load(f)[f"{scene}/{phase}"]


def dataset_from_files(scene, phase):
    """A train/val split re-assembled from the per-file windows (data/files/*.npz), files concatenated like the
    reference's loader concatenates them (utils/dataloader.py:177-236)."""
    obs, pred, sse, base = [], [], [], 0
    for name in split_files(scene, phase):
        o, p, b = _windows(np.load(os.path.join(GOLDEN, "data", "files", name.replace(".txt", ".npz"))))
        obs.append(o)
        pred.append(p)
        sse.append(b.astype(np.int32) + base)
        base += o.shape[0]
    return np.concatenate(obs), np.concatenate(pred), np.concatenate(sse)


def dataset(scene, phase):
    """-> obs (N,8,2) f32, pred (N,12,2) f32, seq_start_end (M,2) int32 (utils/dataloader.py windows)."""
    whole = os.path.join(GOLDEN, "data", f"{scene}_{phase}.npz")
    if os.path.exists(whole):
        return _windows(np.load(whole))
    return dataset_from_files(scene, phase)


def fit_input(scene):
    """A split's train+val with the y-flip augmentation (utils/trainer.py:51-53, utils/utils.py:79-81): what
    the reference fits U and the anchors on."""
    o1, p1, _ = dataset(scene, "train")
    o2, p2, _ = dataset(scene, "val")
    obs, pred = np.concatenate([o1, o2]), np.concatenate([p1, p2])
    flip = np.array([[[1.0, -1.0]]], dtype=np.float32)
    return np.concatenate([obs, obs * flip]), np.concatenate([pred, pred * flip])


def eth_fit_input():
    return fit_input("eth")


def static_dist(scene):
    return float(manifest()["static_dist"][scene])


def sign_align(U, U_ref):
    """Flip columns of U to the sign of U_ref (LAPACK's SVD signs are arbitrary)."""
    s = np.sign((U * U_ref).sum(axis=0))
    s[s == 0] = 1
    return U * s
