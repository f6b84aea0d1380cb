import sys, time, torch
sys.path.insert(0, ".")
from eigentrajectory_amd import ops
from eigentrajectory_amd.synth import gaussian_points_np
dev = torch.device("cuda:0")
for n in (100000, 1000000, 10000000):
    x = torch.from_numpy(gaussian_points_np(6, n, seed=3, n_blobs=7)).to(dev)
    for name, fn in (("exact init", lambda: ops.kmeans_init_farthest(x, 20, 17)), ("reforder init", lambda: ops.kmeans_init_farthest_reference_order(x, 20, 17))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t = time.perf_counter() - t0
        print(n, name, f"{t*1e3:.3f} ms", flush=True)
