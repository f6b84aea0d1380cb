import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, ".")
from eigentrajectory_amd import ops, _lib as L
from eigentrajectory_amd.synth import gaussian_points_np
dev = torch.device("cuda:0")
for n, seed in ((10000, 100), (10000, 101), (100000, 100)):
    x = torch.from_numpy(gaussian_points_np(6, n, seed=seed, n_blobs=0)).to(dev)
    c0 = ops.kmeans_init_farthest_reference_order(x, 20, 17)
    for it in (2, 3):
        L.set_option("reforder_filter_min_lp", 5)
        a = ops.kmeans_fit_reference_order(x, c0, it, -1.0)
        L.set_option("reforder_filter_min_lp", 4)
        b = ops.kmeans_fit_reference_order(x, c0, it, -1.0)
        bad = torch.nonzero(a["labels"] != b["labels"]).flatten()
        print(n, seed, it, "labels differ:", len(bad), "positions mod 1024:", sorted(set((bad % 1024).tolist()))[:20], "groups:", sorted(set((bad // 1024).tolist()))[:12])
    if hasattr(L.lib(), "et_debug_rfcheck"):
        buf = (C.c_uint * 64)()
        L.lib().et_debug_rfcheck(buf)
        print("   wrongly kept so far:", buf[0])
