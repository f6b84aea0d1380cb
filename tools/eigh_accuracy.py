import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from eigentrajectory_amd import ops
from oracle import et_oracle as eo
eo.build(); dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
worst = 0.0; diff_bits = 0; tot = 0
for trial in range(200):
    n = int(rng.choice([4, 8, 16, 24, 33, 64])); k = min(n, 6)
    m = rng.standard_normal((n, 3 * n)) * 10.0 ** rng.uniform(-3, 3)
    if trial % 3 == 0: m[:, : n // 2] *= 1e3
    g = (m @ m.T)
    U, s = ops.eigh_topk(torch.from_numpy(g).to(dev), k)
    Ur, sr = eo.eigh_topk(g, k)
    U = U.cpu().numpy(); s = s.cpu().numpy()
    worst = max(worst, float(np.abs(U - Ur).max()), float(np.abs(s / sr - 1).max()))
    diff_bits += int((U != Ur).sum()); tot += U.size
    # orthonormality and residual in fp64 terms
    assert np.abs(U.T.astype(np.float64) @ U - np.eye(k)).max() < 5e-6
print(os.path.basename(os.environ.get("ET_LIBETAMD", "default")), "max |U - U_oracle|, |s/s_oracle - 1|:", worst, "differing fp32 entries:", diff_bits, "of", tot)
