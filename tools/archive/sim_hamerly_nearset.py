"""CPU estimate (fp64 replay of the bench's fit: synthetic trajectories -> descriptors -> K = 20, farthest-first start) of how
many points a Lloyd iteration could certify WITHOUT READING THEIR COORDINATES, for two bound schemes:
  B  Hamerly: stale upper bound to the own centroid (aged by that centroid's movement), stale lower bound to all others (aged
     by the LARGEST movement of any centroid since the bound was taken), s_a = half the distance to the nearest other centroid;
  C  the lower bound aged only by the movement of the clusters that the centroid-to-centroid test cannot exclude
     (||c_a - c_j|| < 2 u), from a per-cluster history of cumulative movements (K floats per iteration).
Columns: fraction of the points whose coordinates must be read after all ("need-x": the upper bound is tightened with the exact
distance) and fraction that then still needs the scan over all clusters.
Round 5 result at N = 2e5: B is useless on this data (outlier clusters move by 10-100 units per iteration: 65-100 % need-x from
iteration 7 on); C leaves 10-27 % need-x and 3-14 % full scans -- but need-x points are scattered, one 64-byte sector each:
0.15 x 64 B = 9.6 B per point + 6 B of bounds per point is no less than the 15 B per point the packed loop streams today.
    python tools/sim_hamerly_nearset.py [N]"""
import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from eigentrajectory_amd.synth import synthetic_trajectories_np
from oracle import et_oracle as O
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 300000
obs, pred = synthetic_trajectories_np(N, seed=0, min_disp=1e-3)
pn = O.normalize(obs, pred, True).reshape(N, 24).astype(np.float64)
_, V = np.linalg.eigh(pn.T @ pn)
X = np.ascontiguousarray((pn @ V[:, ::-1][:, :6]).astype(np.float32)).astype(np.float64)  # (N, 6)
K = 20
np.random.seed(0)
C = O.kmeans_init_farthest(np.ascontiguousarray(X.T.astype(np.float32)), K, np.random.randint(N))
C = (C[0] if isinstance(C, tuple) else C).astype(np.float64).T  # (K, 6)
def dists(X, C):
    return np.sqrt(np.maximum(((X[:, None, :] - C[None]) ** 2).sum(-1), 0))
T = 100
# state for scheme B (global max drift for l) and C (near-set decay)
lab = None
res = []
cumC = [np.zeros(K)]  # cumulative drift per cluster
for it in range(T):
    if it == 0:
        D = dists(X, C)
        lab = D.argmin(1)
        Ds = np.sort(D, axis=1)
        u = Ds[:, 0].copy(); l = Ds[:, 1].copy(); tr = np.zeros(N, int)  # refresh time
        uB = u.copy(); lB = l.copy(); trB = tr.copy()
        res.append((1.0, 1.0, 1.0, 1.0))
    else:
        cum = cumC[it]
        cc = np.sqrt(((C[:, None] - C[None]) ** 2).sum(-1)); np.fill_diagonal(cc, np.inf)
        s = 0.5 * cc.min(1)
        # ---- scheme B
        ueff = uB + (cum[lab] - cumC_arr[trB, lab])
        gl = (cumC_arr[it] - cumC_arr[trB]).max(1)  # max over j of cum drift since refresh (includes own; conservative)
        leff = lB - gl
        cert1 = ueff < np.maximum(leff, s[lab])
        # tighten u for the others
        idx = np.where(~cert1)[0]
        dn = np.sqrt(((X[idx] - C[lab[idx]]) ** 2).sum(1))
        cert2 = dn < np.maximum(leff[idx], s[lab[idx]])
        uB[idx] = dn; # u refreshed at time it -> need separate refresh times for u and l; approximate: keep tr for l only
        # handle u refresh time separately
        # (for simplicity store u refresh by resetting cumulative reference)
        full = idx[~cert2]
        # ---- scheme C: near-set decay
        ueffC = u + (cum[lab] - cumC_arr[tru, lab])
        dj = cumC_arr[it][None, :] - cumC_arr[trl]          # (N, K) cumulative drift of each cluster since the point's l refresh
        near = cc[lab] < 2 * ueffC[:, None]                   # (N, K) clusters that the centroid-distance test cannot exclude
        near[np.arange(N), lab] = False
        dec = np.where(near, dj, 0).max(1)
        leffC = l - dec
        anynear = near.any(1)
        certC1 = (~anynear) | (ueffC < leffC)
        idxC = np.where(~certC1)[0]
        dnC = np.sqrt(((X[idxC] - C[lab[idxC]]) ** 2).sum(1))
        nearC = cc[lab[idxC]] < 2 * dnC[:, None]
        nearC[np.arange(len(idxC)), lab[idxC]] = False
        decC = np.where(nearC, dj[idxC], 0).max(1)
        certC2 = (~nearC.any(1)) | (dnC < l[idxC] - decC)
        fullC = idxC[~certC2]
        res.append((1 - cert1.mean(), len(full) / N, 1 - certC1.mean(), len(fullC) / N))
        # ground truth assignment + refresh of the failing points (both schemes share labels: exact)
        D = dists(X, C)
        newlab = D.argmin(1)
        assert (newlab[cert1] == lab[cert1]).all() and (newlab[certC1] == lab[certC1]).all()
        chk = np.setdiff1d(idx, full); assert (newlab[chk] == lab[chk]).all()
        chk = np.setdiff1d(idxC, fullC); assert (newlab[chk] == lab[chk]).all()
        Ds = np.sort(D, axis=1)
        # B: refresh
        uB[idx] = dn; truB[idx] = it
        uB[full] = Ds[full, 0]; lB[full] = Ds[full, 1]; trB[full] = it
        # C: refresh
        u[idxC] = dnC; tru[idxC] = it
        u[fullC] = Ds[fullC, 0]; l[fullC] = Ds[fullC, 1]; trl[fullC] = it
        changed = (newlab != lab).mean()
        lab = newlab
        res[-1] = res[-1] + (changed,)
    Cn = np.stack([X[lab == j].mean(0) if (lab == j).any() else C[j] for j in range(K)])
    drift = np.sqrt(((Cn - C) ** 2).sum(1))
    cumC.append(cumC[-1] + drift)
    cumC_arr = np.stack(cumC)
    if it == 0:
        tru = np.zeros(N, int); trl = np.zeros(N, int); truB = np.zeros(N, int)
    # fix B's u aging to use truB
    C = Cn
    if it >= 1:
        pass
print("it: B need-x  B full-scan | C need-x  C full-scan | changed   (max drift)")
for it in range(1, T):
    if it < 12 or it % 8 == 0:
        r = res[it]
        print(f"{it:3d}: {r[0]:.4f} {r[1]:.4f} | {r[2]:.4f} {r[3]:.4f} | {r[4]:.4f}   {(cumC[it+1]-cumC[it]).max():.4f} 2nd {np.sort(cumC[it+1]-cumC[it])[-2]:.4f}")
