"""Exactness check (run on the GPU box): the benchmark's own data distribution (heavy-tailed projected coefficients) at a large N
against the oracle, bit for bit."""
import sys, os, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from eigentrajectory_amd import ops
from eigentrajectory_amd.synth import synthetic_trajectories_torch
from oracle import et_oracle as oracle
oracle.build()
dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
obs, pred = synthetic_trajectories_torch(n, dev, seed=0, min_disp=1e-3)
g_obs, g_pred, _ = ops.fit_gram(obs, pred, 1, 0.0, 1)
(U_obs, _), (U_pred, _) = ops.eigh_topk_batch([g_obs, g_pred], 6)
_, c_pred, _, _ = ops.norm_project(obs, pred, U_obs, U_pred, None, None, 1, want_flag=False)
x = c_pred.contiguous()
xn = x.cpu().numpy()
print("max |x|", np.abs(xn).max(), "median |x|", np.median(np.abs(xn)))
c0 = ops.kmeans_init_farthest(x, 20, 12345)
t0 = time.time(); r0, _ = oracle.kmeans_init_farthest(xn, 20, 12345); print("oracle init", time.time() - t0, "s")
print("init equal:", np.array_equal(c0.cpu().numpy(), r0))
res = ops.kmeans_fit(x, c0, iters, 1e-4)
t0 = time.time(); ref = oracle.kmeans_fit(xn, r0, iters, 1e-4); print("oracle fit", time.time() - t0, "s")
print("n_iter", res["n_iter"], ref["n_iter"], "labels equal:", np.array_equal(res["labels"].cpu().numpy(), ref["labels"]),
      "centroids equal:", np.array_equal(res["centroids"].cpu().numpy(), ref["centroids"]),
      "trace equal:", np.array_equal(res["trace"].cpu().numpy(), ref["trace"]))
