#!/usr/bin/env python3
"""Freeze golden vectors by IMPORTING the reference (never copying it).

Run only in the build container, where /root/reference is mounted:

    python tools/make_golden.py --ref /root/reference --out tests/golden

The reference has no tests of its own, so these captured
inputs/outputs are the pin for the oracle and for the HIP path
(fixtures G1-G9).  Only data is written: inputs and the
reference's outputs on them.  Nothing here is imported by the product.
"""
from __future__ import annotations

import argparse
import contextlib
import io
import json
import os
import re
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

from eigentrajectory_amd.synth import gaussian_points_np, synthetic_trajectories_np  # noqa: E402

SCENES = ["eth", "hotel", "univ", "zara1", "zara2"]


def save(out, name, **arrays):
    path = os.path.join(out, name)
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}  {os.path.getsize(path) / 1024:.1f} KiB")


def quantise(traj_f32):
    q = np.rint(traj_f32.astype(np.float64) * 1e4).astype(np.int32)
    back = (q.astype(np.float64) / 1e4).astype(np.float32)
    assert np.array_equal(back, traj_f32), "dataset is not 4-decimal exact"
    return q


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    os.makedirs(os.path.join(out, "data"), exist_ok=True)

    sys.path.insert(0, args.ref)
    os.chdir(args.ref)  # dataset paths in the reference are relative
    from threadpoolctl import threadpool_limits

    from EigenTrajectory import EigenTrajectory, TrajNorm
    from EigenTrajectory.anchor import ETAnchor
    from EigenTrajectory.descriptor import ETDescriptor
    from EigenTrajectory.kmeans import BatchKMeans
    from utils.dataloader import TrajectoryDataset
    from utils.metrics import compute_batch_ade, compute_batch_fde
    from utils.utils import DotDict, augment_trajectory, get_exp_config

    torch.set_num_threads(1)
    manifest = {"torch": torch.__version__, "numpy": np.__version__}
    import sklearn
    manifest["sklearn"] = sklearn.__version__

    # ------------------------------------------------------------------ data
    print("[data] ETH/UCY windows via the reference's TrajectoryDataset")
    data = {}
    for scene in SCENES:
        phases = ["train", "val", "test"]
        for ph in phases:
            ds = TrajectoryDataset(f"./datasets/{scene}/{ph}/", obs_len=8, pred_len=12)
            full = torch.cat([ds.obs_traj, ds.pred_traj], dim=1).numpy()
            sse = np.asarray(ds.seq_start_end, dtype=np.int32)
            data[(scene, ph)] = (ds.obs_traj.clone(), ds.pred_traj.clone(), sse)
            keep = ph == "test" or scene == "eth"
            if keep:
                save(os.path.join(out, "data"), f"{scene}_{ph}.npz", q=quantise(full), seq_start_end=sse)
            print(f"    {scene}/{ph}: {full.shape[0]} peds, {len(sse)} scenes")

    cfgs = {}
    for scene in SCENES:
        cfg = get_exp_config(f"./config/eigentrajectory-{{baseline}}-{scene}.json")
        cfgs[scene] = cfg
    manifest["static_dist"] = {s: cfgs[s].static_dist for s in SCENES}

    def hyper(scene):
        return cfgs[scene]

    # -------------------------------------------------------------------- G1
    print("[G1] TrajNorm on ETH test")
    obs_t, pred_t, _ = data[("eth", "test")]
    g1 = {}
    for sca in (True, False):
        tn = TrajNorm(ori=True, rot=True, sca=sca)
        tn.calculate_params(obs_t)
        on, pn = tn.normalize(obs_t), tn.normalize(pred_t)
        tag = "sca1" if sca else "sca0"
        g1[f"{tag}_ori"] = tn.traj_ori.numpy()
        g1[f"{tag}_rot"] = tn.traj_rot.numpy()
        if sca:
            g1[f"{tag}_sca"] = tn.traj_sca.numpy()
        g1[f"{tag}_obs_norm"] = on.numpy()
        g1[f"{tag}_pred_norm"] = pn.numpy()
        g1[f"{tag}_pred_roundtrip"] = tn.denormalize(pn).numpy()
    save(out, "g1_trajnorm_eth_test.npz", **g1)

    # -------------------------------------------------------------------- G2
    print("[G2] fit (SVD + sklearn anchors, single-thread) per scene")

    class ZeroStub(torch.nn.Module):
        def __init__(self, k, s):
            super().__init__()
            self.k, self.s = k, s

        def forward(self, x):
            return torch.zeros(self.k, x.size(1), self.s)

    class LinearStub(torch.nn.Module):
        """C_pred_refine[k',n,s] = sum_j W[s,k',j] * cat(C_obs, obs_ori)[j,n]"""

        def __init__(self, w):
            super().__init__()
            self.w = torch.nn.Parameter(w)

        def forward(self, x):
            return torch.einsum("skj,jn->kns", self.w, x)

    hook = DotDict(
        model_forward_pre_hook=lambda obs_data, obs_ori, addl_info=None: torch.cat([obs_data, obs_ori], dim=0),
        model_forward=lambda input_data, baseline_model: baseline_model(input_data),
        model_forward_post_hook=lambda output_data, addl_info=None: output_data,
    )
    w_lin = torch.tensor(np.random.default_rng(123).standard_normal((20, 6, 8)).astype(np.float32) * 0.1)

    fitted = {}
    g2 = {}
    for scene in SCENES:
        hp = hyper(scene)
        obs = torch.cat([data[(scene, "train")][0], data[(scene, "val")][0]], dim=0)
        pred = torch.cat([data[(scene, "train")][1], data[(scene, "val")][1]], dim=0)
        obs, pred = augment_trajectory(obs, pred)
        model = EigenTrajectory(ZeroStub(hp.k, hp.num_samples), hook, hp)
        with threadpool_limits(limits=1):
            model.calculate_parameters(obs, pred)
        fitted[scene] = {k: v.detach().clone() for k, v in model.state_dict().items()}
        mask = (obs[:, -1] - obs[:, -3]).div(2).norm(p=2, dim=-1) > hp.static_dist
        for key, val in fitted[scene].items():
            g2[f"{scene}.{key}"] = val.numpy()
        g2[f"{scene}.n_moving"] = np.int64(mask.sum().item())
        g2[f"{scene}.n_static"] = np.int64((~mask).sum().item())
        # singular values (the reference discards them: descriptor.py:134-135)
        for tag, sel, sca in (("m", mask, True), ("s", ~mask, False)):
            d = ETDescriptor(hp, norm_sca=sca)
            on, pn = d.normalize_trajectory(obs[sel], pred[sel])
            _, s_o, _ = d.truncated_SVD(on, k=16)
            _, s_p, _ = d.truncated_SVD(pn, k=24)
            g2[f"{scene}.sigma_obs_{tag}"] = s_o.numpy()
            g2[f"{scene}.sigma_pred_{tag}"] = s_p.numpy()
        print(f"    {scene}: N={obs.size(0)} moving={int(mask.sum())} static={int((~mask).sum())}")
    save(out, "g2_fit_all_scenes.npz", **g2)

    # -------------------------------------------------------------------- G3
    print("[G3] descriptor_evaluation SVD table, k=1..12, five test splits")
    g3 = {}
    for scene in SCENES:
        obs_t, pred_t, _ = data[(scene, "test")]
        tn = TrajNorm(ori=True, rot=True, sca=False)
        tn.calculate_params(obs_t)
        on, pn = tn.normalize(obs_t), tn.normalize(pred_t)
        n = obs_t.size(0)
        A, B = on.reshape(n, 16).T, pn.reshape(n, 24).T
        U_o, S_o, _ = torch.linalg.svd(A, full_matrices=False)
        U_p, S_p, _ = torch.linalg.svd(B, full_matrices=False)
        errs = np.zeros((12, 2), dtype=np.float64)
        for k in range(1, 13):
            Ao = (U_o[:, :k] @ (U_o[:, :k].T @ A)).T.reshape(n, 8, 2)
            Bo = (U_p[:, :k] @ (U_p[:, :k].T @ B)).T.reshape(n, 12, 2)
            errs[k - 1, 0] = (tn.denormalize(Ao) - obs_t).norm(p=2, dim=-1).mean().item()
            errs[k - 1, 1] = (tn.denormalize(Bo) - pred_t).norm(p=2, dim=-1).mean().item()
        g3[f"{scene}.err"] = errs
        g3[f"{scene}.sigma_obs"] = S_o.numpy()
        g3[f"{scene}.sigma_pred"] = S_p.numpy()
        print(f"    {scene}: k=6 obs {errs[5, 0]:.4f} pred {errs[5, 1]:.4f}")
    save(out, "g3_descriptor_evaluation.npz", **g3)

    # ---------------------------------------------------------------- G4, G5
    print("[G4/G5] projection, anchor+reconstruction fwd/bwd on ETH test with the ETH fit")
    hp = hyper("eth")
    obs_t, pred_t, _ = data[("eth", "test")]
    g45 = {}
    gen = torch.Generator().manual_seed(7)
    for tag, sca in (("m", True), ("s", False)):
        d = ETDescriptor(hp, norm_sca=sca)
        a = ETAnchor(hp)
        d.U_obs_trunc.data = fitted["eth"][f"ET_{tag}_descriptor.U_obs_trunc"].clone()
        d.U_pred_trunc.data = fitted["eth"][f"ET_{tag}_descriptor.U_pred_trunc"].clone()
        a.C_anchor.data = fitted["eth"][f"ET_{tag}_anchor.C_anchor"].clone()
        # sca=True on motionless rows gives inf/NaN; keep only finite rows for the m descriptor
        disp = (obs_t[:, -1] - obs_t[:, -3]).norm(p=2, dim=-1)
        rows = torch.nonzero(disp > 1e-3).squeeze(1) if sca else torch.arange(obs_t.size(0))
        o, p = obs_t[rows], pred_t[rows]
        C_obs, C_pred = d.projection(o, p)
        g45[f"{tag}.rows"] = rows.numpy()
        g45[f"{tag}.C_obs"] = C_obs.numpy()
        g45[f"{tag}.C_pred"] = C_pred.numpy()
        C_ref = torch.randn((hp.k, o.size(0), hp.num_samples), generator=gen, requires_grad=True)
        recon = d.reconstruction(a(C_ref))
        dtraj = torch.randn(recon.shape, generator=gen)
        (recon * dtraj).sum().backward()
        g45[f"{tag}.C_refine"] = C_ref.detach().numpy()
        g45[f"{tag}.recon"] = recon.detach().numpy()
        g45[f"{tag}.dtraj"] = dtraj.numpy()
        g45[f"{tag}.dC"] = C_ref.grad.numpy()
    save(out, "g45_project_reconstruct_eth_test.npz", **g45)

    # -------------------------------------------------------------------- G6
    print("[G6] wrapper forward with stub predictors, per scene batch, five test splits")
    g6 = {"linear_stub_w": w_lin.numpy()}
    summary = {}
    for scene in SCENES:
        hp = hyper(scene)
        obs_t, pred_t, sse = data[(scene, "test")]
        for stub_name, stub in (("zero", ZeroStub(hp.k, hp.num_samples)), ("linear", LinearStub(w_lin.clone()))):
            model = EigenTrajectory(stub, hook, hp)
            sd = dict(fitted[scene])
            for k, v in stub.state_dict().items():
                sd[f"baseline_model.{k}"] = v
            model.load_state_dict(sd)
            model.eval()
            ades, fdes, losses = [], [], []
            with torch.no_grad():
                for (s, e) in sse:
                    o, p = obs_t[s:e], pred_t[s:e]
                    outd = model(o, p)
                    ades.append(compute_batch_ade(outd["recon_traj"], p))
                    fdes.append(compute_batch_fde(outd["recon_traj"], p))
                    losses.append([outd["loss_eigentraj"].item(), outd["loss_euclidean_ade"].item(),
                                   outd["loss_euclidean_fde"].item()])
            ades, fdes = np.concatenate(ades), np.concatenate(fdes)
            g6[f"{scene}.{stub_name}.ade"] = ades.astype(np.float32)
            g6[f"{scene}.{stub_name}.fde"] = fdes.astype(np.float32)
            g6[f"{scene}.{stub_name}.losses"] = np.asarray(losses, dtype=np.float32)
            if scene == "eth":
                g6[f"{scene}.{stub_name}.recon_last"] = outd["recon_traj"].numpy()
            summary[f"{scene}.{stub_name}"] = [float(ades.mean()), float(fdes.mean())]
            print(f"    {scene}/{stub_name}: ADE {ades.mean():.5f} FDE {fdes.mean():.5f}")
    manifest["g6_ade_fde"] = summary
    save(out, "g6_wrapper_stub_predictors.npz", **g6)

    # ---------------------------------------------------------------- G7, G8
    print("[G7/G8] BatchKMeans traces + euc_sim bit patterns")
    g7 = {}

    def replay_history(km, x, c0, n_iter):
        """Centroids entering every Lloyd iteration + the final ones, produced by the reference's own
        get_labels / compute_centroids (kmeans.py:143-198), so each step can be checked in isolation
        (whole-run label equality is ill-conditioned on unstructured data)."""
        c = c0.clone()
        hist = [c[0].numpy().copy()]
        for _ in range(n_iter):
            _, lb = km.get_labels(x, c)
            c = km.compute_centroids(x, lb)
            hist.append(c[0].numpy().copy())
        assert torch.equal(c, km.centroids) or (torch.isnan(c) == torch.isnan(km.centroids)).all()
        return np.stack(hist).astype(np.float32)

    cases = [("gauss", 1000, 0), ("gauss", 10000, 0), ("blobs", 10000, 12), ("gauss", 100000, 0)]
    for kind, n, blobs in cases:
        x_np = gaussian_points_np(6, n, seed=11, n_blobs=blobs)
        x = torch.from_numpy(x_np)[None].contiguous()
        np.random.seed(0)
        first = np.random.randint(n)
        np.random.seed(0)
        km = BatchKMeans(n_clusters=20, n_redo=1, max_iter=100, tol=1e-4, init_mode="kmeans++", verbose=True)
        np.random.seed(0)
        c0 = km.kmeanspp(x)
        np.random.seed(0)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            labels = km.fit(x)
        trace = [(float(m.group(1)), float(m.group(2))) for m in
                 re.finditer(r"error=([-+0-9.eEnaif]+), inertia=([-+0-9.eEnaif]+)", buf.getvalue())]
        hist = replay_history(km, x, c0, len(trace))
        q_np = gaussian_points_np(6, 512, seed=12, n_blobs=blobs)
        pred_labels = km.predict(torch.from_numpy(q_np)[None].contiguous())
        tag = f"{kind}{n}"
        g7[f"{tag}.first_index"] = np.int64(first)
        g7[f"{tag}.blobs"] = np.int64(blobs)
        g7[f"{tag}.c0"] = c0[0].numpy()
        g7[f"{tag}.trace"] = np.asarray(trace, dtype=np.float64)
        g7[f"{tag}.history"] = hist
        g7[f"{tag}.labels"] = labels[0].numpy().astype(np.uint8)
        g7[f"{tag}.centroids"] = km.centroids[0].numpy()
        g7[f"{tag}.query_labels"] = pred_labels[0].numpy().astype(np.uint8)
        print(f"    {tag}: first={first} iters={len(trace)} final error={trace[-1][0]:.3e} inertia={trace[-1][1]:.6f}")
    # k-means on real descriptor coefficients (what anchor generation clusters): ETH moving C_pred
    hp = hyper("eth")
    obs = torch.cat([data[("eth", "train")][0], data[("eth", "val")][0]], dim=0)
    pred = torch.cat([data[("eth", "train")][1], data[("eth", "val")][1]], dim=0)
    obs, pred = augment_trajectory(obs, pred)
    mask = (obs[:, -1] - obs[:, -3]).div(2).norm(p=2, dim=-1) > hp.static_dist
    d = ETDescriptor(hp, norm_sca=True)
    d.U_obs_trunc.data = fitted["eth"]["ET_m_descriptor.U_obs_trunc"].clone()
    d.U_pred_trunc.data = fitted["eth"]["ET_m_descriptor.U_pred_trunc"].clone()
    _, C_pred = d.projection(obs[mask], pred[mask])
    x = C_pred[None].contiguous()
    np.random.seed(0)
    first = np.random.randint(x.size(-1))
    np.random.seed(0)
    km = BatchKMeans(n_clusters=20, verbose=True)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        labels = km.fit(x)
    trace = [(float(m.group(1)), float(m.group(2))) for m in
             re.finditer(r"error=([-+0-9.eEnaif]+), inertia=([-+0-9.eEnaif]+)", buf.getvalue())]
    np.random.seed(0)
    c0_eth = km.kmeanspp(x)
    g7["ethm.c0"] = c0_eth[0].numpy()
    g7["ethm.history"] = replay_history(km, x, c0_eth, len(trace))
    g7["ethm.x"] = C_pred.numpy()
    g7["ethm.first_index"] = np.int64(first)
    g7["ethm.trace"] = np.asarray(trace, dtype=np.float64)
    g7["ethm.labels"] = labels[0].numpy().astype(np.uint8)
    g7["ethm.centroids"] = km.centroids[0].numpy()
    print(f"    ethm: N={x.size(-1)} first={first} iters={len(trace)} inertia={trace[-1][1]:.6f}")
    # empty-cluster / NaN propagation case: 30 points, only 8 distinct -> duplicated farthest-first picks
    base = gaussian_points_np(6, 8, seed=13)
    x_np = np.ascontiguousarray(np.tile(base, (1, 4))[:, :30])
    x = torch.from_numpy(x_np)[None].contiguous()
    np.random.seed(0)
    km = BatchKMeans(n_clusters=20, max_iter=5)
    np.random.seed(0)
    c0 = km.kmeanspp(x)
    np.random.seed(0)
    labels = km.fit(x)
    g7["dup.x"] = x_np
    g7["dup.first_index"] = np.int64(np.random.RandomState(0).randint(30))
    g7["dup.c0"] = c0[0].numpy()
    g7["dup.fit_returned_none"] = np.bool_(labels is None)
    g7["dup.centroids_is_none"] = np.bool_(km.centroids is None)
    ms, lb = km.get_labels(x, c0)
    g7["dup.labels_iter0"] = lb[0].numpy().astype(np.uint8)
    nc = km.compute_centroids(x, lb)
    g7["dup.centroids_iter0"] = nc[0].numpy()
    ms1, lb1 = km.get_labels(x, nc)
    g7["dup.labels_iter1"] = lb1[0].numpy().astype(np.uint8)
    g7["dup.maxsims_iter1_isnan"] = torch.isnan(ms1[0]).numpy()
    print(f"    dup: fit returned None={labels is None}; iter0 NaN centroids={int(torch.isnan(nc).any(dim=1).sum())}")
    save(out, "g7_batchkmeans.npz", **g7)

    a = torch.from_numpy(gaussian_points_np(6, 96, seed=21))
    b = torch.from_numpy(gaussian_points_np(6, 20, seed=22))
    y = BatchKMeans.euc_sim(a, b)
    save(out, "g8_euc_sim.npz", a=a.numpy(), b=b.numpy(), y=y.numpy(),
         a_norm=a.pow(2).sum(dim=-2).numpy(), b_norm=b.pow(2).sum(dim=-2).numpy())

    # -------------------------------------------------------------------- G9
    print("[G9] ADE/FDE on seeded inputs")
    rng = np.random.default_rng(31)
    p = torch.from_numpy(rng.standard_normal((20, 37, 12, 2)).astype(np.float32))
    g = torch.from_numpy(rng.standard_normal((37, 12, 2)).astype(np.float32))
    save(out, "g9_metrics.npz", pred=p.numpy(), gt=g.numpy(),
         ade=compute_batch_ade(p, g), fde=compute_batch_fde(p, g))

    # synthetic-generator pin: the numpy stream must be the same on the GPU box
    o, pr = synthetic_trajectories_np(64, seed=0)
    save(out, "synth_pin.npz", obs=o, pred=pr, pts=gaussian_points_np(6, 64, seed=11))

    with open(os.path.join(out, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("done")


if __name__ == "__main__":
    main()
