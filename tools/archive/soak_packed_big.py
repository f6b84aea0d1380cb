"""Soak at production sizes (run on the GPU box): trace-less fits of shards of 2^21 ... 6e6 points on the packed copy (the
library's default there) against the same fits with the fp32 filter (ET_KMEANS_PACKED=0) -- both exact, so labels,
centroids, iteration count, error and inertia must agree bit for bit.  python tools/soak_packed_big.py [seed] [cases]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from eigentrajectory_amd import _lib as L  # noqa: E402
from eigentrajectory_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
pf = L.lib().et_internal_kmeans_packed_fits
pf.restype = ctypes.c_longlong
bad, t0, packed0 = 0, time.time(), pf()
for case in range(cases):
    rng = np.random.default_rng(seed0 * 7919 + case)
    g = torch.Generator(device=dev)
    g.manual_seed(seed0 * 7919 + case)
    n = int(rng.integers((1 << 21) // 4, 1_500_000)) * 4
    K = int(rng.integers(3, 33))
    nb = int(rng.integers(1, 40))
    x = torch.randn((6, n), generator=g, device=dev)
    centres = torch.randn((6, nb), generator=g, device=dev) * float(rng.uniform(0.5, 8.0))
    x += centres[:, torch.randint(0, nb, (n,), generator=g, device=dev)]
    if rng.random() < 0.5:  # outliers
        idx = torch.randint(0, n, (max(1, n // int(rng.integers(2000, 200000))),), generator=g, device=dev)
        x[:, idx] *= 10.0 ** float(rng.uniform(1, 4))
    if rng.random() < 0.3:  # far from the origin
        x += torch.randn((6, 1), generator=g, device=dev) * 10.0 ** float(rng.uniform(1, 3))
    if rng.random() < 0.3:
        x[int(rng.integers(0, 6))] = 0.0
    x = (x * 10.0 ** float(rng.uniform(-6, 6))).contiguous()
    c0 = ops.kmeans_init_farthest(x, K, int(rng.integers(0, n)))
    iters = int(rng.integers(8, 40))
    L.set_option("kmeans_packed", 1)
    a = ops.kmeans_fit(x, c0, iters, 1e-4 * float(x.var()), trace=False)
    L.set_option("kmeans_packed", 0)
    b = ops.kmeans_fit(x, c0, iters, 1e-4 * float(x.var()), trace=False)
    ok = (a["n_iter"] == b["n_iter"] and torch.equal(a["labels"], b["labels"])
          and np.array_equal(a["centroids"].cpu().numpy(), b["centroids"].cpu().numpy(), equal_nan=True)
          and np.array_equal(np.float32([a["error"], a["inertia"]]), np.float32([b["error"], b["inertia"]]), equal_nan=True))
    if not ok:
        bad += 1
        print("MISMATCH case", case, "n", n, "K", K, "n_iter", a["n_iter"], b["n_iter"], "labels", bool(torch.equal(a["labels"], b["labels"])),
              "centroids", np.array_equal(a["centroids"].cpu().numpy(), b["centroids"].cpu().numpy(), equal_nan=True),
              "error", a["error"], b["error"], "inertia", a["inertia"], b["inertia"], "done", a["done"], b["done"], flush=True)
    del x
print(f"{cases} cases, {bad} mismatches, {pf() - packed0} fits on the packed copy, {time.time() - t0:.0f} s")
