"""CPU estimate for the NEXT step of the Lloyd loop (DESIGN, "what comes next"): certify whole SETS of points as unchanged
without reading them.  At a partition iteration every point gets its certified margin m_n = G_l - max_{j != l} G_j (scaled
units as in packed_assign_body); while the centroids have drifted by less than a budget since then, a point with
m_n > 4 R_n D1 + 2 D2  (D1 = sum over the iterations of max_j |delta q_j|, D2 = sum of max_j | |q_j'|^2 - |q_j|^2 |)
cannot have changed its arg-max and is not looked at; only the "hot" rest is.  This script replays the bench's fit in
fp64 and reports, per epoch length, the fraction of points that stay cold.  python tools/sim_hot_cold.py [N]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigentrajectory_amd.synth import synthetic_trajectories_np  # noqa: E402
from oracle import et_oracle as O  # noqa: E402

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 300000
obs, pred = synthetic_trajectories_np(N, seed=0, min_disp=1e-3)
pn = O.normalize(obs, pred, True).reshape(N, 24).astype(np.float64)
_, V = np.linalg.eigh(pn.T @ pn)
X = np.ascontiguousarray((pn @ V[:, ::-1][:, :6]).T.astype(np.float32)).astype(np.float64)  # (6, N)
K = 20
np.random.seed(0)
C = O.kmeans_init_farthest(X.astype(np.float32), K, np.random.randint(N))
C = (C[0] if isinstance(C, tuple) else C).astype(np.float64)
mu = X.mean(axis=1, keepdims=True)
P = X - mu
R = np.sqrt((P ** 2).sum(0))
hist = []  # per iteration: centred centroids
labels = None
for it in range(60):
    Q = C - mu
    G = 2 * P.T @ Q - (Q ** 2).sum(0)[None, :]
    labels = np.argmax(G, axis=1)
    hist.append((Q.copy(), G[np.arange(N), labels] - np.partition(G, -2, axis=1)[:, -2], labels.copy()))
    Cn = np.stack([X[:, labels == j].mean(axis=1) if (labels == j).any() else C[:, j] for j in range(K)], axis=1)
    C = Cn
print("iteration: max_j |dq_j|, max_j |d|q_j|^2|, changed %")
for t in range(1, len(hist)):
    dq = np.sqrt(((hist[t][0] - hist[t - 1][0]) ** 2).sum(0))
    dn = np.abs((hist[t][0] ** 2).sum(0) - (hist[t - 1][0] ** 2).sum(0))
    if t % 4 == 1 or t < 8:
        print(f"  {t:3d}  {dq.max():9.4f}  {dn.max():10.3f}   {100 * (hist[t][2] != hist[t - 1][2]).mean():6.3f}    (bulk-only max |dq| {np.sort(dq)[-1]:.4f}, 2nd {np.sort(dq)[-2]:.4f})")
for start in (8, 16, 30, 45):
    for length in (4, 8, 12):
        if start + length >= len(hist):
            continue
        m0 = hist[start][1]
        D1 = D2 = 0.0
        for t in range(start + 1, start + length + 1):
            D1 += np.sqrt(((hist[t][0] - hist[t - 1][0]) ** 2).sum(0)).max()
            D2 += np.abs((hist[t][0] ** 2).sum(0) - (hist[t - 1][0] ** 2).sum(0)).max()
        hot = (m0 <= 4 * R * D1 + 2 * D2).mean()
        truly = np.zeros(N, bool)
        for t in range(start + 1, start + length + 1):
            truly |= hist[t][2] != hist[start][2]
        print(f"epoch from iteration {start:2d}, {length:2d} iterations: D1 {D1:8.4f} D2 {D2:9.3f}  hot {100 * hot:6.2f} %   (points that really change label in it: {100 * truly.mean():5.2f} %)")
