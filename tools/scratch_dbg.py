import sys, numpy as np, torch
sys.path.insert(0, ".")
from eigentrajectory_amd import ops, _lib as L
from eigentrajectory_amd.synth import gaussian_points_np
dev = torch.device("cuda:0")
for n in (10000, 100000):
  for seed in (100, 101, 116, 117):
    x = torch.from_numpy(gaussian_points_np(6, n, seed=seed, n_blobs=0 if seed < 116 else 3 + seed - 116)).to(dev)
    c0 = ops.kmeans_init_farthest_reference_order(x, 20, 17)
    for it in (1, 2, 3, 5, 10, 30):
        L.set_option("reforder_filter_min_lp", 5)
        a = ops.kmeans_fit_reference_order(x, c0, it, -1.0)
        L.set_option("reforder_filter_min_lp", 4)
        b = ops.kmeans_fit_reference_order(x, c0, it, -1.0)
        nd = int((a["labels"] != b["labels"]).sum())
        cd = int((a["centroids"] != b["centroids"]).sum())
        print(n, seed, it, "labels differ:", nd, "centroid entries differ:", cd, "inertia", a["inertia"], b["inertia"], flush=True)
        if nd:
            idx = torch.nonzero(a["labels"] != b["labels"]).flatten()[:8].tolist()
            print("   first differing points", idx, [int(a["labels"][i]) for i in idx], [int(b["labels"][i]) for i in idx])
            break
