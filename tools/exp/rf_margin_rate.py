"""Share of points the gap bounds of the reference-order Lloyd loop cannot decide (library built with -DET_EXP_RF_CHECK)."""
import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, ".")
from eigentrajectory_amd import ops, _lib as L
from eigentrajectory_amd.synth import gaussian_points_np, synthetic_trajectories_torch
dev = torch.device("cuda:0")
def counters():
    buf = (C.c_uint * 64)()
    fn = L.lib().et_debug_rfcheck
    rc = fn(buf)
    return list(buf)[:4]
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
which = sys.argv[2] if len(sys.argv) > 2 else "blobs"
if which == "blobs":
    x = torch.from_numpy(gaussian_points_np(6, n, seed=3, n_blobs=7)).to(dev)
else:
    obs, pred = synthetic_trajectories_torch(n, dev, seed=0, min_disp=1e-3)
    g_obs, g_pred, _ = ops.fit_gram(obs, pred, 1, 0.0, 1)
    (U_obs, _), (U_pred, _) = ops.eigh_topk_batch([g_obs, g_pred], 6)
    _, x, _, _ = ops.norm_project(obs, pred, U_obs, U_pred, None, None, 1, want_flag=False)
    x = x.contiguous()
c0 = ops.kmeans_init_farthest(x, 20, 17)
prev = counters()
for iters in (2, 3, 5, 10, 20, 40, 100):
    ops.kmeans_fit_reference_order(x, c0, iters, -1.0, trace=False)
    torch.cuda.synchronize()
    cur = counters()
    d = [a - b for a, b in zip(cur, prev)]
    prev = cur
    print(f"{which} N={n} fit of {iters} iterations: undecided {d[0]} of {d[1]} point-visits = {d[0] / max(d[1], 1):.4f}; groups overflowed {d[2]} of {d[3]}")
