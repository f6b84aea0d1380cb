#!/usr/bin/env python3
"""End-to-end fixture for config 5's data path (EigenTrajectory-AgentFormer, univ split): the reference's wrapper +
its AgentFormer bridge + a seeded, randomly initialised AgentFormerLight, run on CPU in the build container.

    python tools/make_golden_agentformer.py --ref /root/reference --out tests/golden

Stored per scene (three univ test scenes): the pre-hook input the predictor received (`pre_motion`), the predictor's
raw output (`_dec_motion`), and what the reference made of it (`recon_traj`, the three losses, best-of-S ADE / FDE).
A test replays the recorded predictor output through THIS build's wrapper + bridge contract and must land on the
same trajectories and metrics -- the predictor network itself (third-party) is not part of the path.
Only data is written; nothing of the reference is copied.
"""
import argparse
import os
import sys
from collections import defaultdict

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
    from baseline.agentformer import (TrajectoryPredictor, model_forward, model_forward_post_hook,
                                      model_forward_pre_hook)
    from baseline.agentformer.utils.config import Config
    from EigenTrajectory import EigenTrajectory
    from utils.metrics import compute_batch_ade, compute_batch_fde
    from utils.utils import DotDict, get_exp_config

    torch.set_num_threads(1)
    hp = get_exp_config("./config/eigentrajectory-{baseline}-univ.json")
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:  # Config(create_dirs=True) wants a results root it can write to
        cfg = Config("./baseline/agentformer/agentformer_pre.yml", tmp_dir=tmp, create_dirs=False) \
            if "tmp_dir" in Config.__init__.__code__.co_varnames else Config("./baseline/agentformer/agentformer_pre.yml")
    cfg.past_frames, cfg.future_frames = hp.k + 2, hp.k          # utils/trainer.py:388-391
    cfg.motion_dim, cfg.forecast_dim = 1, hp.num_samples
    cfg.input_type, cfg.pred_type, cfg.sn_out_type, cfg.scene_orig_all_past = ['pos'], 'pos', None, False
    cfg.nz, cfg.ar_train, cfg.learn_prior = 0, False, False
    torch.manual_seed(2024)
    predictor = TrajectoryPredictor(cfg).eval()

    captured = {}

    def forward_and_capture(input_data, baseline_model):
        captured["pre_motion"] = input_data["pre_motion"].detach().clone()
        out = model_forward(input_data, baseline_model)
        captured["dec_motion"] = out["_dec_motion"].detach().clone()
        return out

    hook = DotDict(model_forward_pre_hook=model_forward_pre_hook, model_forward=forward_and_capture,
                   model_forward_post_hook=model_forward_post_hook)
    model = EigenTrajectory(predictor, hook, hp).eval()
    g2 = G.load("g2_fit_all_scenes.npz")
    sd = model.state_dict()
    for key in list(sd):
        if key.startswith("ET_"):
            sd[key] = torch.from_numpy(g2[f"univ.{key}"])
    model.load_state_dict(sd)

    obs, pred, sse = G.dataset("univ", "test")
    out = {"static_dist": np.float32(hp.static_dist)}
    picks = [i for i, (s, e) in enumerate(sse) if 5 <= e - s <= 30][:3]
    for j, i in enumerate(picks):
        s, e = sse[i]
        o, p = torch.from_numpy(obs[s:e]), torch.from_numpy(pred[s:e])
        with torch.no_grad():
            res = model(o, p)
        out[f"scene{j}.index"] = np.int64(i)
        out[f"scene{j}.pre_motion"] = captured["pre_motion"].numpy()
        out[f"scene{j}.dec_motion"] = captured["dec_motion"].numpy()
        out[f"scene{j}.recon_traj"] = res["recon_traj"].numpy()
        out[f"scene{j}.losses"] = np.asarray([res["loss_eigentraj"].item(), res["loss_euclidean_ade"].item(),
                                              res["loss_euclidean_fde"].item()], np.float32)
        out[f"scene{j}.ade"] = np.asarray(compute_batch_ade(res["recon_traj"], p), np.float32)
        out[f"scene{j}.fde"] = np.asarray(compute_batch_fde(res["recon_traj"], p), np.float32)
        print(f"  scene {i}: N={e - s} pre_motion {tuple(captured['pre_motion'].shape)} dec_motion "
              f"{tuple(captured['dec_motion'].shape)} ADE {out[f'scene{j}.ade'].mean():.4f}")
    path = os.path.join(args.out, "g12_agentformer_univ.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
