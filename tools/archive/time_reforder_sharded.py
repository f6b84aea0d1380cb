"""Cost of the sharded form of the reference-order Lloyd iteration on ONE GPU (one rank: the record is copied where
ncclAllGather would run): per-iteration time from the difference of two fits that never stop (tol < 0), against the
single-GPU loop of the same library.  What it shows is the price of the extra kernel boundary + the serial level 3 in a
single workgroup; a real N > 1 run adds RCCL's all-gather latency per iteration on top.
    python tools/time_reforder_sharded.py [N ...]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from eigentrajectory_amd import ops
from eigentrajectory_amd.synth import gaussian_points_np


def wall(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    dev = torch.device("cuda:0")
    sizes = [int(float(a)) for a in sys.argv[1:]] or [100000, 1000000, 10000000]
    for n in sizes:
        x = torch.from_numpy(gaussian_points_np(6, n, seed=1, n_blobs=12)).to(dev)
        c0 = ops.kmeans_init_farthest_reference_order(x, 20, 7)
        out = {}
        for name, fit in [("single", lambda it: ops.kmeans_fit_reference_order(x, c0, it, -1.0, trace=False)),
                          ("sharded_1rank", lambda it: ops.kmeans_fit_reference_order_sharded(x, c0, [n], 0, None, it, -1.0, trace=False)),
                          ("sharded_as_rank0_of_8", None)]:
            if fit is None:  # rank 0's own work in an 8-rank split, its record copied: levels 0 .. 2 of 1/8 of the points
                s8 = ops.reference_order_shard_sizes(n, 8)
                xl = x[:, :s8[0]].contiguous()
                fit = lambda it: ops.kmeans_fit_reference_order_sharded(xl, c0, [s8[0]], 0, None, it, -1.0, trace=False)
            fit(8)
            t8, t40 = wall(lambda: fit(8)), wall(lambda: fit(40))
            out[name] = (t40 - t8) / 32 * 1e6
        a = ops.kmeans_fit_reference_order(x, c0, 6, -1.0)
        b = ops.kmeans_fit_reference_order_sharded(x, c0, [n], 0, None, 6, -1.0)
        same = bool(torch.equal(a["centroids"], b["centroids"]) and torch.equal(a["labels"], b["labels"]))
        print(f"N={n}: us/iteration " + "  ".join(f"{k} {v:.1f}" for k, v in out.items()) + f"  same_bits={same}", flush=True)


if __name__ == "__main__":
    main()
