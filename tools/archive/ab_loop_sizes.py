#!/usr/bin/env python3
"""Lloyd loop time per shard size (development aid): run once per loop form, e.g.
    ET_OPT_KMEANS_LOOP=chain python tools/ab_loop_sizes.py ; ET_OPT_KMEANS_LOOP=persist python tools/ab_loop_sizes.py
prints, per N, the wall time of et_kmeans_fit (100 iterations, no trace; median of 7) and the time per iteration."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigentrajectory_amd import ops
from eigentrajectory_amd.synth import synthetic_trajectories_torch

dev = torch.device("cuda:0")
sizes = [int(float(a)) for a in sys.argv[1:]] or [20_000, 70_000, 100_000, 300_000, 600_000, 1_000_000, 2_000_000, 4_000_000, 10_000_000]
print("loop form:", os.environ.get("ET_OPT_KMEANS_LOOP", "default"))
for n in sizes:
    obs, pred = synthetic_trajectories_torch(n, dev, seed=0, min_disp=1e-3)
    g_obs, g_pred, _ = ops.fit_gram(obs, pred, 1, 0.0, 1)
    (U_obs, _), (U_pred, _) = ops.eigh_topk_batch([g_obs, g_pred], 6)
    _, c_pred, _, _ = ops.norm_project(obs, pred, U_obs, U_pred, None, None, 1, want_flag=False)
    x = c_pred.contiguous()
    del obs, pred
    c0 = ops.kmeans_init_farthest(x, 20, 12345 % n)
    ws = ops.kmeans_workspace(n, 6, 20, dev)
    ts = []
    for r in range(9):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = ops.kmeans_fit(x, c0, 100, 1e-4, workspace=ws, trace=False)
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts[2:]))
    print(f"N {n:9d}: fit {t*1e3:7.3f} ms  {t*1e6/res['n_iter']:6.2f} us/iteration ({res['n_iter']} iterations)", flush=True)
