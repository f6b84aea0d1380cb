#!/usr/bin/env python3
"""SVD block of the reference's script/descriptor_evaluation.py (:87-112) on the HIP kernels:
reconstruction error of the rank-k descriptor, k = 1..12, per ETH/UCY test split.

    python scripts/descriptor_evaluation.py --data tests/golden/data        # committed fixtures
    python scripts/descriptor_evaluation.py --raw <dir with eth/test/*.txt ...>

The curve-fitting baselines of that script (Linear / Bezier / B-spline, 100 000 Adam steps each) are a
paper table, not part of the descriptor path, and are not reproduced."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from eigentrajectory_amd import ops  # noqa: E402


def svd_table(obs, pred, ks=range(1, 13)):
    """-> (len(ks), 2) mean L2 errors (obs, pred) with TrajNorm(ori, rot, sca=False), like the reference."""
    g_obs, g_pred, _ = ops.fit_gram(obs, pred, ops.MODE_STATIC, which=0)
    out = []
    for k in ks:
        (U_obs, _), (U_pred, _) = ops.eigh_topk_batch([g_obs, g_pred], k)
        c_obs, c_pred, nrm, _ = ops.norm_project(obs, pred, None, None, U_obs, U_pred, ops.MODE_STATIC, want_flag=False)
        r_obs = ops.anchor_reconstruct(c_obs.unsqueeze(-1), None, None, None, U_obs, ops.MODE_STATIC, nrm=nrm)[0]
        r_pred = ops.anchor_reconstruct(c_pred.unsqueeze(-1), None, None, None, U_pred, ops.MODE_STATIC, nrm=nrm)[0]
        out.append([(r_obs - obs).norm(p=2, dim=-1).mean().item(), (r_pred - pred).norm(p=2, dim=-1).mean().item()])
    return np.asarray(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default=None, help="directory with <scene>_test.npz fixtures (tests/golden/data)")
    ap.add_argument("--raw", default=None, help="dataset root with <scene>/test/*.txt")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for scene in ["eth", "hotel", "univ", "zara1", "zara2"]:
        if args.raw:
            from eigentrajectory_amd.data import TrajectoryData
            d = TrajectoryData(os.path.join(args.raw, scene, "test"))
            obs, pred = d.obs_traj.to(dev), d.pred_traj.to(dev)
        else:
            root = args.data or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
            z = np.load(os.path.join(root, f"{scene}_test.npz"))
            full = torch.from_numpy((z["q"].astype(np.float64) / 1e4).astype(np.float32)).to(dev)
            obs, pred = full[:, :8].contiguous(), full[:, 8:].contiguous()
        print(f"=== {scene} ({obs.shape[0]} pedestrians) ===Singular Value Decomposition===")
        for k, (eo, ep) in zip(range(1, 13), svd_table(obs, pred)):
            print(f"k: {k}\tnum params: {k}\tobs error: {eo:.4f}\tpred error: {ep:.4f}")


if __name__ == "__main__":
    main()
