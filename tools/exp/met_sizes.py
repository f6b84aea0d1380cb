import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from eigentrajectory_amd import ops
from eigentrajectory_amd.synth import synthetic_trajectories_torch
dev = torch.device("cuda:0")
S = 20
def med(fn, reps=12):
    for _ in range(3): fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
for n in (200_000, 500_000, 1_000_000, 2_000_000, 5_000_000, 10_000_000):
    obs, pred = synthetic_trajectories_torch(n, dev, seed=0, min_disp=1e-3)
    g_obs, g_pred, _ = ops.fit_gram(obs, pred, ops.MODE_MOVING, 0.0, 1)
    (Uo, _), (Up, _) = ops.eigh_topk_batch([g_obs, g_pred], 6)
    _, _, nrm, _, pose = ops.norm_project(obs, pred, Uo, Up, None, None, ops.MODE_MOVING, want_flag=False, want_pose=True)
    C20 = torch.randn((6, n, S), device=dev) * 0.1
    A = torch.randn((6, S), device=dev)
    t = med(lambda: ops.anchor_reconstruct_metrics(C20, pred, A, None, Up, None, ops.MODE_MOVING, pose=pose))
    print("N %9d  %.4f ms  %.1f ns per 1000 trajectories  %.3f of 8 TB/s  (inputs %.0f MB)" % (n, t, t * 1e6 / n * 1000 / 1000, 604 * n / t / 1e6 / 8000, 600 * n / 1e6))
    del C20, obs, pred, nrm, pose
    torch.cuda.empty_cache()
