"""Randomized soak of the reference-order k-means (fast form, csrc/et_kmeans_reforder.hip namespace fast) against the oracle's
literal restatement: random N (every leftover of the cascade: partial chunk / group / block, N mod 4, N mod 32), K in 1..32,
batches of 1..3 problems, scales over 16 decades, outliers, duplicated points, far-from-origin data, with and without the
matrix-core certification.  Labels, centroid bits, per-iteration errors, iteration count.   python tools/soak_reforder.py [cases] [seed]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from eigentrajectory_amd import ops, _lib as L
from oracle import et_oracle as eo

eo.build()
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    n = int(rng.choice([1024, 1025, 1027, 1056, 2047, 4096, 4100, 5003, 16384, 16389, 20001, 33000, 65536 + 37, 70001, 131072 + 5, 150001]))
    if rng.rand() < 0.5:
        n = int(rng.randint(1024, 60000))
    K = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 12, 20, 20, 20, 31, 32]))
    l = int(rng.choice([1, 1, 2, 3]))
    iters = int(rng.randint(2, 9)) if rng.rand() < 0.7 else int(rng.randint(9, 40))  # (some long fits)
    scale = float(10.0 ** rng.uniform(-8, 8)) if rng.rand() < 0.3 else 1.0
    xs = []
    for b in range(l):
        kind = rng.randint(5)
        x = rng.standard_normal((6, n)).astype(np.float32)
        if kind == 1:
            x += (rng.standard_normal((6, 1)) * 50).astype(np.float32)          # far from the origin
        elif kind == 2:
            x[:, :: int(rng.randint(7, 200))] *= np.float32(rng.choice([30.0, 1e3]))  # outliers
        elif kind == 3:
            cen = rng.standard_normal((6, int(rng.randint(2, 12)))).astype(np.float32) * 4
            x = cen[:, rng.randint(cen.shape[1], size=n)] + x * np.float32(0.3)  # blobs
        elif kind == 4:
            x[:, n // 2:] = x[:, : n - n // 2]                                    # every point twice: exact ties
        xs.append((x * np.float32(scale)).astype(np.float32))
    xs = np.stack(xs)
    firsts = [int(rng.randint(n)) for _ in range(l)]
    c0 = np.stack([eo.kmeans_init_farthest(xs[b], K, firsts[b], reference_order=True)[0] for b in range(l)])
    for skip_min in (0, 1 << 40):  # the farthest-first picks, with and without the big-shard point skip
        L.set_option("reforder_init_skip_min", skip_min)
        for b in range(l):
            got = ops.kmeans_init_farthest_reference_order(torch.from_numpy(xs[b]).to(dev), K, firsts[b]).cpu().numpy()
            if not np.array_equal(got, c0[b], equal_nan=True):
                bad += 1
                print(f"MISMATCH (farthest-first) case {case}: n={n} K={K} scale={scale:g} skip_min={skip_min}", flush=True)
    L.set_option("reforder_init_skip_min", 1 << 21)
    ref = eo.kmeans_fit_batch_reference_order(list(xs), list(c0), iters, 1e-4)
    for flt in (9, 4):
        L.set_option("reforder_filter_min_lp", flt)
        L.set_option("reforder_single_update", 1 if flt == 9 else int(rng.randint(2)))  # (both forms of the update kernel)
        runs = ops.kmeans_fit_reference_order_batch(torch.from_numpy(xs).to(dev), torch.from_numpy(c0).to(dev), iters, 1e-4)
        ok = True
        for b, r in enumerate(runs):
            ok &= r["n_iter"] == ref["n_iter"]
            ok &= bool(np.array_equal(r["labels"].cpu().numpy(), ref["labels"][b]))
            ok &= bool(np.array_equal(r["centroids"].cpu().numpy(), ref["centroids"][b], equal_nan=True))
            ok &= bool(np.array_equal(r["trace"].cpu().numpy()[:, 0], ref["trace"][:, 0], equal_nan=True))
        if not ok:
            bad += 1
            print(f"MISMATCH case {case}: n={n} K={K} l={l} iters={iters} scale={scale:g} filter_lp={flt}", flush=True)
    if case % 20 == 19:
        print(f"{case + 1} cases, {bad} mismatches", flush=True)
L.set_option("reforder_filter_min_lp", 9)
L.set_option("reforder_single_update", 1)
print(f"done: {cases} cases x 2 forms, {bad} mismatches")
