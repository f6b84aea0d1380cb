#!/usr/bin/env python3
"""Whole-run equality of BatchKMeans with the imported reference over the 96 runs of tests/golden/g7c_batchkmeans_seeds.npz
(tools/make_golden_batchkmeans_seeds.py), for both summation modes -- the table DESIGN.md 4 quotes.

    python tools/g7c_rate.py            # the CPU oracle (oracle/et_oracle.c), ~10 min
    python tools/g7c_rate.py --gpu      # the product (libetamd.so) on cuda:0
"""
import argparse
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eigentrajectory_amd.synth import gaussian_points_np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--sizes", default="1000,10000,100000")
    args = ap.parse_args()
    z = np.load(os.path.join(ROOT, "tests", "golden", "g7c_batchkmeans_seeds.npz"))
    K = int(z["K"])
    if args.gpu:
        import torch
        from eigentrajectory_amd import ops
        dev = torch.device("cuda:0")

        def run(x, first, mode):
            xt = torch.from_numpy(x).to(dev)
            if mode == "exact":
                c0 = ops.kmeans_init_farthest(xt, K, first)
                r = ops.kmeans_fit(xt, c0, 100, 1e-4, trace=False)
            else:
                c0 = ops.kmeans_init_farthest_reference_order(xt, K, first)
                r = ops.kmeans_fit_reference_order(xt, c0, 100, 1e-4, trace=False)
            return c0.cpu().numpy(), r["labels"].cpu().numpy(), r["n_iter"], r["centroids"].cpu().numpy()
    else:
        from oracle import et_oracle as eo
        eo.build()

        def run(x, first, mode):
            ref = mode != "exact"
            c0, _ = eo.kmeans_init_farthest(x, K, first, reference_order=ref)
            r = eo.kmeans_fit(x, c0, 100, 1e-4, sums=mode)
            return c0, r["labels"], r["n_iter"], r["centroids"]
    table = {}
    t0 = time.time()
    for n in [int(v) for v in args.sizes.split(",")]:
        for seed in z["seeds"]:
            tag = f"n{n}.s{seed}"
            blobs = int(z[tag + ".blobs"])
            x = gaussian_points_np(6, n, seed=int(seed), n_blobs=blobs)
            for mode in ("exact", "reference-order"):
                c0, labels, n_iter, cen = run(x, int(z[tag + ".first_index"]), mode)
                same_c0 = np.array_equal(c0, z[tag + ".c0"])
                same = (hashlib.sha256(labels.astype(np.uint8).tobytes()).digest() == bytes(z[tag + ".labels_sha256"])
                        and n_iter == int(z[tag + ".n_iter"]))
                bits = np.array_equal(cen, z[tag + ".centroids"])
                key = (n, "gauss" if blobs == 0 else "blobs", mode)
                t = table.setdefault(key, [0, 0, 0, 0])
                t[0] += same_c0
                t[1] += same
                t[2] += bits
                t[3] += 1
                if not same:
                    print(f"  {tag} {mode}: labels differ (iterations {n_iter} / {int(z[tag + '.n_iter'])})", flush=True)
        print(f"N = {n} done ({time.time() - t0:.0f} s)", flush=True)
    print("N, data, mode: initial centroids equal / whole run (labels + iteration count) equal / final centroids bit-equal / runs")
    for key, t in sorted(table.items()):
        print(key, t)


if __name__ == "__main__":
    main()
