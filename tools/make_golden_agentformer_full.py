#!/usr/bin/env python3
"""G15: config 5's data path at the extent of G14 -- EigenTrajectory-AgentFormer inference on the univ split, 20 samples,
a fixed tenth of univ's test scenes (scene index % 10 == 0: 95 of 947 scenes, the same ones G14 records for SGCN), run through
the reference's wrapper + its AgentFormer bridge + a seeded, randomly initialised AgentFormerLight on CPU in the build container.

    python tools/make_golden_agentformer_full.py --ref /root/reference --out tests/golden

Stored, concatenated over the recorded scenes in scene order: the scene indices and sizes, the pre-hook input the predictor
received (`pre_motion`, (k + 2, N, 1) per scene), the predictor's raw output (`_dec_motion`), the reference's best-of-20
ADE / FDE per pedestrian (utils/trainer.py:173-195: the inference form `model(obs)`, metrics against pred) and their means.
A GPU test and an oracle test replay the recorded predictor outputs through THIS build's wrapper + bridge and must land on
the same per-pedestrian and split-level ADE / FDE.  Only data is written; nothing of the reference is copied."""
import argparse
import os
import sys
import tempfile
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
                    help="G15b: ALL 947 test scenes of univ (24 334 pedestrians) -> g15b_agentformer_univ_all.npz")
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
    with tempfile.TemporaryDirectory() as tmp:
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
    picks = [i for i in range(len(sse)) if args.univ_all or i % 10 == 0]
    pres, decs, ades, fdes, sizes = [], [], [], [], []
    t0 = time.time()
    pre_shape = dec_shape = None
    for i in picks:
        s, e = sse[i]
        o, p = torch.from_numpy(obs[s:e]), torch.from_numpy(pred[s:e])
        with torch.no_grad():
            res = model(o)  # the test loop's call (utils/trainer.py:183)
        n = e - s
        pre, dec = captured["pre_motion"].numpy(), captured["dec_motion"].numpy()
        if pre_shape is None:
            pre_shape, dec_shape = pre.shape, dec.shape
            print("first scene: N =", n, "pre_motion", pre.shape, "dec_motion", dec.shape)
        pres.append(pre)
        decs.append(dec)
        ades.append(np.asarray(compute_batch_ade(res["recon_traj"], p), np.float32))
        fdes.append(np.asarray(compute_batch_fde(res["recon_traj"], p), np.float32))
        sizes.append(n)
    ade, fde = np.concatenate(ades), np.concatenate(fdes)
    print(f"univ: {len(picks)} scenes, {sum(sizes)} pedestrians, ADE {ade.mean():.5f} FDE {fde.mean():.5f}  ({time.time() - t0:.0f} s)")
    # the axis that carries the pedestrians: the one whose length is the scene's N in both recordings
    n0 = sizes[0]
    pre_axis = [a for a, d in enumerate(pre_shape) if d == n0][0]
    dec_axis = [a for a, d in enumerate(dec_shape) if d == n0 or d == n0 * hp.num_samples][0]
    out = {"static_dist": np.float32(hp.static_dist), "scene_index": np.asarray(picks, np.int64),
           "scene_size": np.asarray(sizes, np.int64), "pre_axis": np.int64(pre_axis), "dec_axis": np.int64(dec_axis),
           "dec_per_pedestrian": np.int64(dec_shape[dec_axis] // n0),
           "pre_motion": np.concatenate(pres, axis=pre_axis).astype(np.float32),
           "dec_motion": np.concatenate(decs, axis=dec_axis).astype(np.float32),
           "ade": ade, "fde": fde,
           "ade_fde_mean": np.asarray([ade.mean(dtype=np.float64), fde.mean(dtype=np.float64)])}
    path = os.path.join(args.out, "g15b_agentformer_univ_all.npz" if args.univ_all else "g15_agentformer_univ_tenth.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
