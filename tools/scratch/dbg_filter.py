import sys, os
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from eigentrajectory_amd import ops
from oracle import et_oracle as oracle
import _golden as G
z = G.load("g7_batchkmeans.npz")
tag = "gauss10000"
from eigentrajectory_amd.synth import gaussian_points_np
x = gaussian_points_np(6, 10000, seed=11, n_blobs=int(z[f"{tag}.blobs"]))
dev = torch.device("cuda:0")
xt = torch.from_numpy(x).to(dev)
c0 = z[f"{tag}.c0"]
sh = ops.KMeansShard(xt, 20); sh.scan()
cen = torch.from_numpy(c0).to(dev).clone(); sh.begin(x.shape[1], cen)
for it in range(40):
    cprev = cen.cpu().numpy().copy()
    part = sh.assign(cen)
    lab = sh.labels().cpu().numpy()
    rl, rm = oracle.kmeans_assign(x, cprev)
    bad = np.nonzero(lab != rl)[0]
    print(it, "mismatches", len(bad))
    if len(bad):
        for n in bad[:5]:
            a = x[:, n].astype(np.float64); 
            sims = 2 * a @ cprev.astype(np.float64) - (a * a).sum() - (cprev.astype(np.float64) ** 2).sum(0)
            o = np.argsort(-sims)
            print("  n", n, "gpu", lab[n], "ref", rl[n], "top sims", o[:3], sims[o[:3]], "|x|", np.sqrt((a*a).sum()))
        break
    sh.update(part, cen, 1e-4)
