"""Exactness soak (run on the GPU box): long randomized soak of the k-means kernels against the oracle (bit-exact)."""
import sys, os, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from eigentrajectory_amd import ops
from oracle import et_oracle as oracle
oracle.build()
dev = torch.device("cuda:0")
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 500
big = len(sys.argv) > 3
traceless = os.environ.get("ET_SOAK_TRACELESS") == "1"  # the trace-less fit (no per-iteration inertia) has its own certification
import ctypes
from eigentrajectory_amd import _lib as L
_pf = L.lib().et_internal_kmeans_packed_fits
_pf.restype = ctypes.c_longlong
packed0 = _pf()  # fits that iterated on the packed copy (ET_OPT_KMEANS_PACKED_MIN=1024 ET_OPT_KMEANS_LOOP=chain ET_SOAK_TRACELESS=1: all of them)
bad = 0
t0 = time.time()
for case in range(cases):
    rng = np.random.default_rng(seed0 * 100003 + case)
    n = int(rng.integers(256, 60000 if big else 8000)) * 4
    K = int(rng.integers(3, 33))
    x = rng.standard_normal((6, n))
    nb = int(rng.integers(1, 12))
    x += rng.standard_normal((6, nb))[:, rng.integers(0, nb, size=n)] * rng.uniform(0.5, 8.0)
    if rng.random() < 0.4:
        idx = rng.choice(n, max(1, n // int(rng.integers(20, 400))), replace=False)
        x[:, idx] *= 10.0 ** rng.uniform(1, 4)
    if rng.random() < 0.3:
        src = rng.integers(0, n, size=n // 3)
        x[:, rng.integers(0, n, size=n // 3)] = x[:, src]
    if rng.random() < 0.3:
        x[int(rng.integers(0, 6))] = 0.0
    if rng.random() < 0.3:
        x[1] = x[0] * 1.5 + 1e-4 * rng.standard_normal(n)
    x = np.ascontiguousarray((x * 10.0 ** rng.uniform(-8, 8)).astype(np.float32))
    first = int(rng.integers(0, n))
    xt = torch.from_numpy(x).to(dev)
    c0 = ops.kmeans_init_farthest(xt, K, first)
    r0, _ = oracle.kmeans_init_farthest(x, K, first)
    ok = np.array_equal(c0.cpu().numpy(), r0, equal_nan=True)
    it = int(rng.integers(5, 40))
    res = ops.kmeans_fit(xt, c0, it, 1e-4, trace=not traceless)
    ref = oracle.kmeans_fit(x, r0, it, 1e-4)
    ok = ok and res["n_iter"] == ref["n_iter"] and np.array_equal(res["labels"].cpu().numpy(), ref["labels"]) \
        and np.array_equal(res["centroids"].cpu().numpy(), ref["centroids"], equal_nan=True)
    if traceless:
        last = ref["trace"][ref["n_iter"] - 1]
        ok = ok and np.array_equal(np.float32([res["error"], res["inertia"]]), np.float32(last), equal_nan=True)
    else:
        ok = ok and np.array_equal(res["trace"].cpu().numpy(), ref["trace"], equal_nan=True)
    if not ok:
        bad += 1
        print("MISMATCH case", case, "n", n, "K", K, flush=True)
print(f"{cases} cases, {bad} mismatches, {_pf() - packed0} fits on the packed copy, {time.time() - t0:.0f} s")
