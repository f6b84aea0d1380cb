import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from eigentrajectory_amd import ops
from eigentrajectory_amd.synth import synthetic_trajectories_torch
from oracle import et_oracle as eo
eo.build()
dev = torch.device("cuda:0")
obs, pred = synthetic_trajectories_torch(1_000_000, dev, seed=0, min_disp=1e-3)
g_obs, g_pred, _ = ops.fit_gram(obs, pred, 1, 0.0, 1)
def t(fn, reps=20):
    for _ in range(3): fn()
    ts=[]
    for _ in range(reps):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))*1e3
U,s = ops.eigh_topk(g_pred, 6)
Ur, sr = eo.eigh_topk(g_pred.cpu().numpy(), 6)
print(os.environ.get("ET_LIBETAMD","default"), "bit-exact vs oracle:", np.array_equal(U.cpu().numpy(), Ur),
      "eigh24 us", round(t(lambda: ops.eigh_topk(g_pred, 6)),1), "eigh16 us", round(t(lambda: ops.eigh_topk(g_obs, 6)),1),
      "batch(16,24) us", round(t(lambda: ops.eigh_topk_batch([g_obs, g_pred], 6)),1))
