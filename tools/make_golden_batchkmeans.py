#!/usr/bin/env python3
"""G7b: the reference's BatchKMeans on a BATCH of l = 3 problems of different difficulty (kmeans.py:228-240: one
error summed over the whole batch stops all problems together).  Run in the build container only:

    python tools/make_golden_batchkmeans.py --ref /root/reference --out tests/golden

Imports the reference (never copies it); writes data only: the inputs, the initial centroids its kmeanspp drew, the
per-iteration (error, inertia) trace it printed, its final labels and centroids."""
from __future__ import annotations

import argparse
import contextlib
import io
import os
import re
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
from eigentrajectory_amd.synth import gaussian_points_np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    sys.path.insert(0, args.ref)
    from EigenTrajectory.kmeans import BatchKMeans
    torch.set_num_threads(1)
    n, K = 4000, 8
    # problem 0 settles in a few iterations (well separated blobs), problem 1 is slow (unstructured), problem 2 has a
    # heavy tail: alone they would stop at different iterations
    xs = np.stack([gaussian_points_np(6, n, seed=81, n_blobs=8) * np.float32(3.0),
                   gaussian_points_np(6, n, seed=82, n_blobs=0),
                   gaussian_points_np(6, n, seed=83, n_blobs=5)])
    xs[2][:, ::97] *= np.float32(25.0)
    x = torch.from_numpy(np.ascontiguousarray(xs))
    out = {"x": xs, "K": np.int64(K)}
    np.random.seed(0)
    out["first_index"] = np.int64(np.random.randint(n))
    km = BatchKMeans(n_clusters=K, n_redo=1, max_iter=100, tol=1e-4, init_mode="kmeans++", verbose=True)
    np.random.seed(0)
    c0 = km.kmeanspp(x)
    out["c0"] = c0.numpy()
    np.random.seed(0)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        labels = km.fit(x)
    trace = [(float(m.group(1)), float(m.group(2))) for m in
             re.finditer(r"error=([-+0-9.eEnaif]+), inertia=([-+0-9.eEnaif]+)", buf.getvalue())]
    out["trace"] = np.asarray(trace, dtype=np.float64)
    out["labels"] = labels.numpy().astype(np.uint8)
    out["centroids"] = km.centroids.numpy()
    # what each problem does ALONE (own stop): the iteration counts differ -- that is what the joint stop must not do
    alone = []
    for b in range(3):
        kb = BatchKMeans(n_clusters=K, n_redo=1, max_iter=100, tol=1e-4, init_mode="kmeans++", verbose=True)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            kb.fit(x[b:b + 1].contiguous(), centroids=c0[b:b + 1].clone())
        alone.append(len(re.findall(r"----iteration", buf.getvalue())))
    out["iterations_alone"] = np.asarray(alone, dtype=np.int64)
    path = os.path.join(args.out, "g7b_batchkmeans_joint_stop.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: joint iterations {len(trace)}, alone {alone}, {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
