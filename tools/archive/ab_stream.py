#!/usr/bin/env python3
"""HISTORICAL (round 4): the streaming kernels this script compares were taken out of libetamd.so in round 5
(tools/lost_forms/project_reconstruct_stream.hip.txt; results: profiles/r04a_stream_ab.txt) -- it runs against the library of
commit 22948e6 only.

Same-process A/B of the projection / S=1 reconstruction kernels at N = 1e7: tile kernels (ET_STREAM=0) against the
streaming kernels with 1..4 workgroups per CU (ET_STREAM_WGS).  HIP-event medians of 30 launches, alternating."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigentrajectory_amd import ops  # noqa: E402
from eigentrajectory_amd.synth import synthetic_trajectories_torch  # noqa: E402

dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
obs, pred = synthetic_trajectories_torch(n, dev, seed=0, min_disp=1e-3)
g_obs, g_pred, _ = ops.fit_gram(obs, pred, ops.MODE_MOVING, 0.0, 1)
(Uo, _), (Up, _) = ops.eigh_topk_batch([g_obs, g_pred], 6)
_, c_pred, nrm, _ = ops.norm_project(obs, pred, Uo, Up, None, None, ops.MODE_MOVING, want_flag=False)
cp = c_pred.view(6, n, 1)


def med(fn, reps=30):
    for _ in range(3):
        fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


configs = [("tile", "0", "")] + [(f"stream wgs={w}", "1", str(w)) for w in (1, 2, 3, 4)]
for rnd in range(2):
    for name, st, wgs in configs:
        os.environ["ET_STREAM"] = st
        if wgs:
            os.environ["ET_STREAM_WGS"] = wgs
        else:
            os.environ.pop("ET_STREAM_WGS", None)
        tp = med(lambda: ops.norm_project(obs, pred, Uo, Up, None, None, ops.MODE_MOVING, want_flag=False))
        tr = med(lambda: ops.anchor_reconstruct(cp, None, None, Up, None, ops.MODE_MOVING, nrm=nrm))
        print(f"round {rnd} {name:14s} project {tp:.4f} ms ({208 * n / tp / 1e6 / 8000:.3f})  reconstruct {tr:.4f} ms "
              f"({136 * n / tr / 1e6 / 8000:.3f})  sum {tp + tr:.4f} ({344 * n / (tp + tr) / 1e6 / 8000:.3f})", flush=True)
