"""Grid-size sweep of the chained Lloyd kernel on small shards (verdict r4 item 5): per-iteration time of a 100-iteration
trace-less fit for workgroup caps x threads per workgroup, same process, alternating.  python tools/ab_loop_grid.py [sizes]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from eigentrajectory_amd import ops, _lib as L
from eigentrajectory_amd.synth import synthetic_trajectories_torch

dev = torch.device("cuda:0")
sizes = [int(float(a)) for a in sys.argv[1:]] or [100000, 300000, 1000000]
for n in sizes:
    o, p = synthetic_trajectories_torch(n, dev, seed=0, min_disp=1e-3)
    g_obs, g_pred, _ = ops.fit_gram(o, p, ops.MODE_MOVING, 0.0, 1)
    (U_obs, _), (U_pred, _) = ops.eigh_topk_batch([g_obs, g_pred], 6)
    _, x, _, _ = ops.norm_project(o, p, U_obs, U_pred, None, None, ops.MODE_MOVING, want_flag=False, want_nrm=False)
    c0 = ops.kmeans_init_farthest(x, 20, 12345 % n)
    L.set_option("kmeans_loop", "chain")
    rows = []
    for threads in (0, 256, 512, 768, 1024):
        for cap in (0, 256, 128, 64, 32, 16):
            L.set_option("kmeans_filter_threads", threads)
            L.set_option("kmeans_loop_grid", cap)
            ts = []
            for rep in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = ops.kmeans_fit(x, c0, 100, 1e-4, trace=False)
                e1.record()
                torch.cuda.synchronize()
                if rep:
                    ts.append(e0.elapsed_time(e1))
            rows.append((min(ts) * 10, threads, cap, r["n_iter"]))
    L.set_option("kmeans_filter_threads", 0)
    L.set_option("kmeans_loop_grid", 0)
    L.set_option("kmeans_loop", "auto")
    base = [t for t, th, cap, _ in rows if th == 0 and cap == 0][0]
    print(f"N = {n}: default {base:.2f} us per iteration; (threads, workgroup cap) -> us per iteration of a 100-iteration fit")
    for t, th, cap, it in sorted(rows)[:10]:
        print(f"   threads {th or 'auto':>5} cap {cap or 'none':>5}: {t:6.2f} us  ({it} iterations)")
