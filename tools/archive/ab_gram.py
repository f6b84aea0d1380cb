"""Same-box timing of the Gram pass (et_fit_gram) and the whole fit at N = 1e7 (ET_LIBETAMD selects a variant build)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from eigentrajectory_amd import ops
from eigentrajectory_amd.synth import synthetic_trajectories_torch
dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
obs, pred = synthetic_trajectories_torch(n, dev, seed=0, min_disp=1e-3)
def t(fn, reps=9):
    fn(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
g = ops.fit_gram(obs, pred, ops.MODE_MOVING, which=1)
gram = t(lambda: ops.fit_gram(obs, pred, ops.MODE_MOVING, which=1))
def fit():
    go, gp, _ = ops.fit_gram(obs, pred, ops.MODE_MOVING, which=1)
    return ops.eigh_topk_batch([go, gp], 6)
full = t(fit)
print(os.path.basename(os.environ.get("ET_LIBETAMD", "default")), f"N={n:.0e} checksum {float(g[0].abs().sum()):.9e} {float(g[1].abs().sum()):.9e}",
      f"gram {gram*1e3:.1f} us ({160*n/gram/1e6:.0f} GB/s)  gram+eigh {full*1e3:.1f} us")
