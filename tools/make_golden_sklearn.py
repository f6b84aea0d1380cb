#!/usr/bin/env python3
"""Pin for the sklearn half of the anchor fit (EigenTrajectory/anchor.py:65-71): outputs of scikit-learn itself.

    python tools/make_golden_sklearn.py --out tests/golden

scikit-learn is the reference's third-party dependency for anchor generation and is not part of
/root/reference; this script runs the installed sklearn (version recorded in the fixture) on inputs
that the tests can regenerate (tests/golden/g7 `ethm.x` = the ETH moving coefficients, and seeded
synthetic point sets) and stores what it produced:

  <tag>.seeds      (10,20) indices drawn by sklearn.cluster.kmeans_plusplus over ten consecutive
                   initialisations sharing ONE RandomState(0), on the mean-centred data -- the draws
                   KMeans(init='k-means++', n_init=10, random_state=0).fit makes
  <tag>.centers    KMeans(...).fit(X).cluster_centers_.T   (d,K)     [single-threaded]
  <tag>.inertia, <tag>.n_iter, <tag>.mean, <tag>.tol

Only data is written.
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
from eigentrajectory_amd.synth import gaussian_points_np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    import sklearn
    from sklearn.cluster import KMeans, kmeans_plusplus
    from threadpoolctl import threadpool_limits

    g7 = np.load(os.path.join(args.out, "g7_batchkmeans.npz"))
    cases = {"ethm": g7["ethm.x"],
             "blobs20000": gaussian_points_np(6, 20000, seed=11, n_blobs=12),
             "gauss50000": gaussian_points_np(6, 50000, seed=11)}
    out = {"sklearn_version": np.array(sklearn.__version__)}
    for tag, C in cases.items():
        X = np.ascontiguousarray(C.T.astype(np.float32))  # (N,d), what anchor.py:65 hands to sklearn
        Xc = X - X.mean(axis=0)
        rs = np.random.RandomState(0)
        seeds = np.stack([kmeans_plusplus(Xc, 20, random_state=rs)[1] for _ in range(10)])
        with threadpool_limits(limits=1):
            km = KMeans(n_clusters=20, random_state=0, init="k-means++", n_init=10).fit(X)
        out[f"{tag}.seeds"] = seeds.astype(np.int64)
        out[f"{tag}.centers"] = np.ascontiguousarray(km.cluster_centers_.T.astype(np.float32))
        out[f"{tag}.inertia"] = np.float64(km.inertia_)
        out[f"{tag}.n_iter"] = np.int64(km.n_iter_)
        out[f"{tag}.mean"] = X.mean(axis=0)
        out[f"{tag}.tol"] = np.float32(km._tol)  # sklearn's own: mean(var(X)) * 1e-4 of the data as given
        print(f"  {tag}: N={X.shape[0]} inertia={km.inertia_:.4f} n_iter={km.n_iter_}")
    # an input that MUST produce empty clusters: 15 distinct locations, 8 copies each, K = 20 -- sklearn re-seeds every empty
    # cluster with the point farthest from its centre (_relocate_empty_clusters_dense) and ends with 5 duplicated centres
    import warnings
    rng = np.random.default_rng(5)
    pts = (rng.standard_normal((6, 15)) * 3).astype(np.float32)
    C = np.ascontiguousarray(np.repeat(pts, 8, axis=1)[:, rng.permutation(120)])
    with threadpool_limits(limits=1), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        km = KMeans(n_clusters=20, random_state=0, init="k-means++", n_init=10).fit(np.ascontiguousarray(C.T))
    out["dup15.x"] = C
    out["dup15.centers"] = np.ascontiguousarray(km.cluster_centers_.T.astype(np.float32))
    out["dup15.inertia"] = np.float64(km.inertia_)
    out["dup15.labels"] = km.labels_.astype(np.int64)
    print(f"  dup15: inertia={km.inertia_:.3e} n_iter={km.n_iter_} clusters used {len(set(km.labels_.tolist()))}")
    path = os.path.join(args.out, "g11_sklearn_anchors.npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {path} {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
