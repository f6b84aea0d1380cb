#!/usr/bin/env python3
"""G14: config 3 at full extent -- EigenTrajectory-SGCN inference, 20 samples, EVERY test scene of eth / hotel / zara1 /
zara2 and a fixed tenth of univ's (scene index % 10 == 0: 95 of 947 scenes; all of univ would be 12 MB of network
outputs), run through the reference's wrapper + its SGCN bridge + a seeded, randomly initialised SGCN on CPU in the build
container (same patching of `.cuda()` as tools/make_golden_sgcn.py).

    python tools/make_golden_sgcn_full.py --ref /root/reference --out tests/golden

Stored per split, concatenated over the recorded scenes in scene order: the scene indices and sizes, the pre-hook input the
network received (`v`, (k + 2) values per pedestrian), the network's raw output ((k, S) values per pedestrian: what the
post-hook turns into C_pred_refine), the reference's best-of-20 ADE / FDE per pedestrian (utils/trainer.py:173-195: the
inference form `model(obs)`, metrics against pred) and their means over the recorded pedestrians.  A GPU test replays the
recorded network outputs through THIS build's wrapper + bridge and must land on the same per-pedestrian and split-level
ADE / FDE.  Only data is written; nothing of the reference is copied."""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--univ-all", action="store_true",
                    help="G14b: ALL 947 test scenes of univ (24 334 pedestrians, utils/trainer.py:173-195's whole loop) -> "
                         "g14b_sgcn_univ_all.npz (12 MB of network outputs), nothing else")
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
    t0 = time.time()
    for scene in (["univ"] if args.univ_all else G.SCENES):
        hp = get_exp_config(f"./config/eigentrajectory-{{baseline}}-{scene}.json")
        torch.manual_seed(1234)
        predictor = TrajectoryPredictor(number_asymmetric_conv_layer=7, embedding_dims=64, number_gcn_layers=1, dropout=0,
                                        obs_len=hp.k + 2, pred_len=hp.k, n_tcn=5, in_dims=1,
                                        out_dims=hp.num_samples).eval()  # utils/trainer.py:288-290
        captured = {}

        def forward_and_capture(input_data, baseline_model):
            v, eyes = input_data
            captured["v"] = v.detach().clone()
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
        picks = [i for i in range(len(sse)) if scene != "univ" or args.univ_all or i % 10 == 0]
        vs, nets, ades, fdes, sizes = [], [], [], [], []
        for i in picks:
            s, e = sse[i]
            o, p = torch.from_numpy(obs[s:e]), torch.from_numpy(pred[s:e])
            with torch.no_grad():
                res = model(o)  # the test loop's call (utils/trainer.py:183)
            n = e - s
            v, net = captured["v"].numpy(), captured["net_out"].numpy()
            assert v.shape == (1, hp.k + 2, n, 1) and net.shape == (hp.k, n, hp.num_samples), (v.shape, net.shape)
            vs.append(v.reshape(hp.k + 2, n))
            nets.append(net)
            ades.append(np.asarray(compute_batch_ade(res["recon_traj"], p), np.float32))
            fdes.append(np.asarray(compute_batch_fde(res["recon_traj"], p), np.float32))
            sizes.append(n)
        print(f"{scene}: {len(picks)} scenes, {sum(sizes)} pedestrians, ADE {np.concatenate(ades).mean():.5f} "
              f"FDE {np.concatenate(fdes).mean():.5f}  ({time.time() - t0:.0f} s)", flush=True)
        out[f"{scene}.static_dist"] = np.float32(hp.static_dist)
        out[f"{scene}.scene_index"] = np.asarray(picks, np.int64)
        out[f"{scene}.scene_size"] = np.asarray(sizes, np.int64)
        out[f"{scene}.v"] = np.concatenate(vs, axis=1).astype(np.float32)            # (k + 2, sum N)
        out[f"{scene}.net_out"] = np.concatenate(nets, axis=1).astype(np.float32)    # (k, sum N, S)
        out[f"{scene}.ade"] = np.concatenate(ades)
        out[f"{scene}.fde"] = np.concatenate(fdes)
        out[f"{scene}.ade_fde_mean"] = np.asarray([np.concatenate(ades).mean(dtype=np.float64), np.concatenate(fdes).mean(dtype=np.float64)])
    path = os.path.join(args.out, "g14b_sgcn_univ_all.npz" if args.univ_all else "g14_sgcn_full_splits.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
