#!/usr/bin/env python3
"""G13: end-to-end fixture for config 3's data path (EigenTrajectory-SGCN inference, 20 samples, all five ETH/UCY
splits): the reference's wrapper + its SGCN bridge + a seeded, randomly initialised SGCN, run on CPU in the build
container.

    python tools/make_golden_sgcn.py --ref /root/reference --out tests/golden

The reference's SGCN moves itself to the GPU inside its constructor and its forward (`.cuda()`, `device='cuda'`,
baseline/sgcn/model.py:53-54,79); there is no GPU here, so for the duration of this script `Tensor.cuda` /
`Module.cuda` are the identity and `torch.zeros_like(..., device='cuda')` stays on the CPU -- the arithmetic is the
reference's own.  Stored per scene (three test scenes per split): the pre-hook input the network received (graph `v`
and the two identity stacks), the network's raw output, and what the reference made of it (`recon_traj`, the three
losses, best-of-S ADE / FDE).  A test replays the recorded network output through THIS build's wrapper + bridge contract
and must land on the same trajectories and metrics -- the network itself (third-party) is not part of the
path.  Only data is written; nothing of the reference is copied."""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    from tests import _golden as G
    sys.path.insert(0, args.ref)
    os.chdir(args.ref)

    # no GPU in the build container: keep the reference's SGCN on the CPU
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    _zeros_like = torch.zeros_like

    def zeros_like_cpu(x, *a, **k):
        k.pop("device", None)
        return _zeros_like(x, *a, **k)
    torch.zeros_like = zeros_like_cpu

    from baseline.sgcn import TrajectoryPredictor, model_forward, model_forward_post_hook, model_forward_pre_hook
    from EigenTrajectory import EigenTrajectory
    from utils.metrics import compute_batch_ade, compute_batch_fde
    from utils.utils import DotDict, get_exp_config

    torch.set_num_threads(1)
    g2 = G.load("g2_fit_all_scenes.npz")
    out = {}
    for scene in G.SCENES:
        hp = get_exp_config(f"./config/eigentrajectory-{{baseline}}-{scene}.json")
        torch.manual_seed(1234)
        predictor = TrajectoryPredictor(number_asymmetric_conv_layer=7, embedding_dims=64, number_gcn_layers=1, dropout=0,
                                        obs_len=hp.k + 2, pred_len=hp.k, n_tcn=5, in_dims=1,
                                        out_dims=hp.num_samples).eval()  # utils/trainer.py:288-290
        captured = {}

        def forward_and_capture(input_data, baseline_model):
            v, eyes = input_data
            captured["v"] = v.detach().clone()
            captured["eye_shapes"] = np.asarray([list(e.shape) for e in eyes], np.int64)
            assert all(torch.equal(e, torch.eye(e.size(-1)).expand_as(e)) for e in eyes)
            res = model_forward(input_data, baseline_model)
            captured["net_out"] = res.detach().clone()
            return res

        hook = DotDict(model_forward_pre_hook=model_forward_pre_hook, model_forward=forward_and_capture,
                       model_forward_post_hook=model_forward_post_hook)
        model = EigenTrajectory(predictor, hook, hp).eval()
        sd = model.state_dict()
        for key in list(sd):
            if key.startswith("ET_"):
                sd[key] = torch.from_numpy(g2[f"{scene}.{key}"])
        model.load_state_dict(sd)
        obs, pred, sse = G.dataset(scene, "test")
        out[f"{scene}.static_dist"] = np.float32(hp.static_dist)
        picks = [i for i, (s, e) in enumerate(sse) if 3 <= e - s <= 40][:3]
        for j, i in enumerate(picks):
            s, e = sse[i]
            o, p = torch.from_numpy(obs[s:e]), torch.from_numpy(pred[s:e])
            with torch.no_grad():
                res = model(o, p)
            tag = f"{scene}.scene{j}"
            out[f"{tag}.index"] = np.int64(i)
            out[f"{tag}.v"] = captured["v"].numpy()
            out[f"{tag}.eye_shapes"] = captured["eye_shapes"]
            out[f"{tag}.net_out"] = captured["net_out"].numpy()
            out[f"{tag}.recon_traj"] = res["recon_traj"].numpy()
            out[f"{tag}.losses"] = np.asarray([res["loss_eigentraj"].item(), res["loss_euclidean_ade"].item(),
                                               res["loss_euclidean_fde"].item()], np.float32)
            out[f"{tag}.ade"] = np.asarray(compute_batch_ade(res["recon_traj"], p), np.float32)
            out[f"{tag}.fde"] = np.asarray(compute_batch_fde(res["recon_traj"], p), np.float32)
            print(f"  {scene} scene {i}: N={e - s} v {tuple(captured['v'].shape)} net_out {tuple(captured['net_out'].shape)} "
                  f"ADE {out[f'{tag}.ade'].mean():.4f} FDE {out[f'{tag}.fde'].mean():.4f}")
    path = os.path.join(args.out, "g13_sgcn_all_scenes.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
