"""Same-box A/B of the streaming descriptor kernels at N = 1e7 (project obs+pred, reconstruct S = 1)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from eigentrajectory_amd import ops
from eigentrajectory_amd.synth import synthetic_trajectories_torch
dev = torch.device("cuda:0")
n = 10_000_000
obs, pred = synthetic_trajectories_torch(n, dev, seed=0, min_disp=1e-3)
g = torch.Generator(device=dev).manual_seed(1)
Uo = torch.linalg.qr(torch.randn((16, 6), device=dev, generator=g))[0].contiguous()
Up = torch.linalg.qr(torch.randn((24, 6), device=dev, generator=g))[0].contiguous()
def t(fn, reps=9):
    fn(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
with torch.no_grad():
    c_obs, c_pred, nrm, _ = ops.norm_project(obs, pred, Uo, Up, None, None, 1, want_flag=False)
    pj = t(lambda: ops.norm_project(obs, pred, Uo, Up, None, None, 1, want_flag=False))
    rc = t(lambda: ops.anchor_reconstruct(c_pred.view(6, n, 1), None, None, Up, None, 1, nrm=nrm))
    gm = t(lambda: ops.fit_gram(obs, pred, 1, 0.0, 1))
print(os.path.basename(os.environ.get("ET_LIBETAMD", "default")),
      f"project {pj:.4f} ms {208*n/pj/1e6:.0f} GB/s | reconstruct S=1 {rc:.4f} ms {136*n/rc/1e6:.0f} GB/s | gram {gm:.4f} ms {160*n/gm/1e6:.0f} GB/s")
