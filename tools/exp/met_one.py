import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from eigentrajectory_amd import ops
from eigentrajectory_amd.synth import synthetic_trajectories_torch
dev = torch.device("cuda:0")
S, n = 20, 2_000_000
obs, pred = synthetic_trajectories_torch(n, dev, seed=0, min_disp=1e-3)
g_obs, g_pred, _ = ops.fit_gram(obs, pred, ops.MODE_MOVING, 0.0, 1)
(Uo, _), (Up, _) = ops.eigh_topk_batch([g_obs, g_pred], 6)
_, _, nrm, _, pose = ops.norm_project(obs, pred, Uo, Up, None, None, ops.MODE_MOVING, want_flag=False, want_pose=True)
C20 = torch.randn((6, n, S), device=dev) * 0.1
A = torch.randn((6, S), device=dev)
for _ in range(5):
    ops.anchor_reconstruct_metrics(C20, pred, A, None, Up, None, ops.MODE_MOVING, pose=pose)
torch.cuda.synchronize()
