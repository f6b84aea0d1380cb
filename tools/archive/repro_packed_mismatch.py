"""Reproduce one case of tools/soak_packed_big.py and locate the first iteration at which the packed and the fp32 fits differ;
compare both with the traced fit and (optionally) the CPU oracle.  python tools/repro_packed_mismatch.py <seed> <case> [oracle]"""
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from eigentrajectory_amd import ops  # noqa: E402
from eigentrajectory_amd import _lib as L  # noqa: E402

dev = torch.device("cuda:0")
seed0, case = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed0 * 7919 + case)
g = torch.Generator(device=dev)
g.manual_seed(seed0 * 7919 + case)
n = int(rng.integers((1 << 21) // 4, 1_500_000)) * 4
K = int(rng.integers(3, 33))
nb = int(rng.integers(1, 40))
x = torch.randn((6, n), generator=g, device=dev)
centres = torch.randn((6, nb), generator=g, device=dev) * float(rng.uniform(0.5, 8.0))
x += centres[:, torch.randint(0, nb, (n,), generator=g, device=dev)]
flags = []
if rng.random() < 0.5:
    idx = torch.randint(0, n, (max(1, n // int(rng.integers(2000, 200000))),), generator=g, device=dev)
    x[:, idx] *= 10.0 ** float(rng.uniform(1, 4))
    flags.append("outliers")
if rng.random() < 0.3:
    x += torch.randn((6, 1), generator=g, device=dev) * 10.0 ** float(rng.uniform(1, 3))
    flags.append("offset")
if rng.random() < 0.3:
    x[int(rng.integers(0, 6))] = 0.0
    flags.append("zero row")
x = (x * 10.0 ** float(rng.uniform(-6, 6))).contiguous()
c0 = ops.kmeans_init_farthest(x, K, int(rng.integers(0, n)))
iters = int(rng.integers(8, 40))
tol = 1e-4 * float(x.var())
print("n", n, "K", K, "iters", iters, flags, "max|x|", float(x.abs().max()), "mean", x.mean(dim=1).cpu().numpy(), "tol", tol)


def fit(packed, it, trace=False):
    L.set_option("kmeans_packed", 1 if packed else 0)
    return ops.kmeans_fit(x, c0, it, tol, trace=trace)


first = None
for it in range(1, iters + 1):
    a, b = fit(True, it), fit(False, it)
    same = torch.equal(a["labels"], b["labels"]) and np.array_equal(a["centroids"].cpu().numpy(), b["centroids"].cpu().numpy(), equal_nan=True)
    if not same:
        first = it
        t = fit(False, it, trace=True)
        nd = int((a["labels"] != b["labels"]).sum())
        print(f"first difference with max_iter = {it}: {nd} labels differ; packed == traced: {torch.equal(a['labels'], t['labels'])}, "
              f"fp32 == traced: {torch.equal(b['labels'], t['labels'])}; n_iter {a['n_iter']} {b['n_iter']} {t['n_iter']}")
        where = torch.nonzero(a["labels"] != b["labels"]).flatten()[:5]
        for w in where.tolist():
            p = x[:, w].double()
            # exact similarities to the centroids the LAST assignment used (the fits return the updated ones: use it - 1's)
            prev = fit(False, it - 1)["centroids"].double() if it > 1 else c0.double()
            y = 2 * (p[:, None] * prev).sum(0) - (prev ** 2).sum(0) - (p ** 2).sum()
            top = torch.topk(y, 2)
            print("  point", w, "labels packed/fp32/traced", int(a["labels"][w]), int(b["labels"][w]), int(t["labels"][w]),
                  "fp64 top-2", top.indices.tolist(), "gap", float(top.values[0] - top.values[1]), "rel", float((top.values[0] - top.values[1]) / top.values[0].abs()))
        break
if first is None:
    print("no difference up to", iters)
if len(sys.argv) > 3 and first is not None:
    from oracle import et_oracle as oracle
    oracle.build()
    ref = oracle.kmeans_fit(x.cpu().numpy(), c0.cpu().numpy(), first, tol)
    a, b = fit(True, first), fit(False, first)
    print("oracle == packed:", np.array_equal(a["labels"].cpu().numpy(), ref["labels"]), " oracle == fp32:", np.array_equal(b["labels"].cpu().numpy(), ref["labels"]))
