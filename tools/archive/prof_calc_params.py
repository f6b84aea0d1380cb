#!/usr/bin/env python3
"""Where calculate_parameters (model.py:34-56) spends its time at the size of the reference's fit sets (development aid)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigentrajectory_amd import EigenTrajectory, ops
import eigentrajectory_amd.anchor as A
from eigentrajectory_amd.synth import synthetic_trajectories_torch
from eigentrajectory_amd.utils import DotDict, default_hyper_params

dev = torch.device("cuda:0")
hooks = DotDict(model_forward_pre_hook=lambda c, o, a=None: torch.cat([c, o], dim=0), model_forward=lambda x, m: m(x),
                model_forward_post_hook=lambda y, a=None: y)
model = EigenTrajectory(torch.nn.Identity(), hooks, default_hyper_params(static_dist=0.3)).to(dev)
o, p = synthetic_trajectories_torch(70_316, dev, seed=6)
for _ in range(3):
    model.calculate_parameters(o, p)
torch.cuda.synchronize()
t = time.perf_counter


def ph(name, f, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = t(); r = f(); torch.cuda.synchronize(); ts.append(t() - t0)
    print(f"{name:44s} {np.median(ts) * 1e3:7.3f} ms")
    return r


sd = 0.3
ph("calculate_parameters", lambda: model.calculate_parameters(o, p))
g = ph("4 grams", lambda: [x for which in (1, 0) for x in ops.fit_gram(o, p, ops.MODE_SPLIT, sd, which)[:2]])
ph("eigh batch (4 matrices)", lambda: ops.eigh_topk_batch(g, 6))
_, U_pred_m, _, U_pred_s = model._U()
r = ph("project", lambda: ops.norm_project(o, p, None, U_pred_m, None, U_pred_s, ops.MODE_SPLIT, sd, want_nrm=False, want_obs=False))
C_pred, flag = r[1], r[3]
moving = flag.bool()
C_m, C_s = ph("split (boolean masks)", lambda: (C_pred[:, moving].contiguous(), C_pred[:, ~moving].contiguous()))
print("moving", tuple(C_m.shape), "static", tuple(C_s.shape))
for nm, C in (("moving", C_m), ("static", C_s)):
    ph(f"sklearn_style_kmeans {nm} (batched)", lambda: A.sklearn_style_kmeans(C, 20))
    ph(f"sklearn_style_kmeans {nm} (one by one)", lambda: A.sklearn_style_kmeans(C, 20, concurrent=False), reps=2)
    X, mean, tol = ph("  center_columns", lambda: ops.center_columns(C))
    U = torch.from_numpy(A.seeding_uniforms(np.random.RandomState(0), 20, 10)).to(dev)
    c0, idx = ph("  seed batch (10)", lambda: ops.kmeanspp_seed_batch(X, 20, U))
    res = ph("  fit batch (10)", lambda: ops.kmeans_fit_batch(X, c0, 300, float(tol.item())))
    print("    iterations", res["n_iter"])
    ph("  predict batch (10)", lambda: ops.kmeans_predict(X, res["centroids"]))
    one = ph("  one fit (longest)", lambda: ops.kmeans_fit(X, c0[int(np.argmax(res["n_iter"]))], 300, float(tol.item()), trace=False))
