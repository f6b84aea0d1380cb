"""Randomized soak of the SHARDED reference-order k-means (et_internal_kmeans_reforder_sharded_run: the loop behind
et_kmeans_fit_reforder_sharded, with the tests' barrier + copy all-gather where ncclAllGather would run) against the
single-GPU reference-order fit: random N over all three level steps' neighbourhoods, 1..5 shards cut at random whole level-2
blocks (empty ranks included), random K, data kinds of tools/soak_reforder.py.  Centroid bits, labels, error trace,
iteration count on EVERY shard.      python tools/soak_reforder_sharded.py [cases] [seed]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from eigentrajectory_amd import ops, _lib as L
from tests._gpu_common import _reference_order_shards_native

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    r = rng.rand()
    if r < 0.6:
        n = int(rng.randint(1024, 200000))
    elif r < 0.8:
        n = int(rng.choice([16384, 32768, 49152, 16384 * 3 + 1, 16384 * 5 - 1, 65536 + 3, 16384 * 7 + 1023]))
    else:
        n = int(rng.randint((4 << 19) + 4, 2600000))  # level step 32: blocks of 131 072 points
    K = int(rng.choice([2, 3, 5, 8, 20, 20, 31, 32]))
    iters = int(rng.randint(2, 10))
    block = int(L.lib().et_kmeans_reforder_shard_block(L.i64(n), 6, K))
    nblocks = -(-n // block)
    P = int(rng.randint(1, 6))
    # random whole-block shard sizes, in order; the remainder of the array goes to the last non-empty shard
    cuts = np.sort(rng.randint(0, nblocks + 1, size=P - 1)) if P > 1 else np.array([], dtype=int)
    edges = np.concatenate([[0], cuts, [nblocks]]) * block
    sizes = [int(min(edges[i + 1], n) - min(edges[i], n)) for i in range(P)]
    assert sum(sizes) == n
    kind = rng.randint(4)
    x = rng.standard_normal((6, n)).astype(np.float32)
    if kind == 1:
        x += (rng.standard_normal((6, 1)) * 50).astype(np.float32)
    elif kind == 2:
        x[:, :: int(rng.randint(7, 200))] *= np.float32(30.0)
    elif kind == 3:
        cen = rng.standard_normal((6, int(rng.randint(2, 12)))).astype(np.float32) * 4
        x = cen[:, rng.randint(cen.shape[1], size=n)] + x * np.float32(0.3)
    xt = torch.from_numpy(x).to(dev)
    c0 = ops.kmeans_init_farthest_reference_order(xt, K, int(rng.randint(n)))
    whole = ops.kmeans_fit_reference_order(xt, c0, iters, 1e-4)
    shards = _reference_order_shards_native(dev, x, c0.cpu().numpy(), sizes, iters, 1e-4)
    ok = np.array_equal(np.concatenate([s["labels"] for s in shards]), whole["labels"].cpu().numpy())
    for s in shards:
        ok &= np.array_equal(s["centroids"], whole["centroids"].cpu().numpy(), equal_nan=True)
        ok &= int(s["state"].iter) == whole["n_iter"]
        ok &= np.array_equal(s["trace"][:, 0], whole["trace"].cpu().numpy()[:, 0], equal_nan=True)
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: n={n} K={K} iters={iters} sizes={sizes}", flush=True)
    if case % 10 == 9:
        print(f"{case + 1} cases, {bad} mismatches", flush=True)
print(f"done: {cases} cases, {bad} mismatches")
