#!/usr/bin/env python3
"""Per-kernel micro-benchmarks on the GPU box (development aid, not part of the product)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigentrajectory_amd import ops
from eigentrajectory_amd.synth import synthetic_trajectories_torch

dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
what = sys.argv[2] if len(sys.argv) > 2 else "all"
obs, pred = synthetic_trajectories_torch(n, dev, seed=0, min_disp=1e-3)

def timeit(fn, reps=10, warm=2, inner=4):
    """median / min time of one call; `inner` back-to-back calls per measurement keep the GPU busy so that
    the host-side launch latency is not counted."""
    for _ in range(warm): fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); a.record()
        for _ in range(inner): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / inner)
    return float(np.median(ts)), float(np.min(ts))

g_obs, g_pred, _ = ops.fit_gram(obs, pred, 1, 0.0, 1)
U_obs, _ = ops.eigh_topk(g_obs, 6); U_pred, _ = ops.eigh_topk(g_pred, 6)
c_obs, c_pred, nrm, _ = ops.norm_project(obs, pred, U_obs, U_pred, None, None, 1, want_flag=False)
if what in ("all", "desc"):
    for name, fn, byt in [
        ("fit_gram", lambda: ops.fit_gram(obs, pred, 1, 0.0, 1), 160),
        ("eigh24", lambda: ops.eigh_topk(g_pred, 6), 0),
        ("project obs+pred", lambda: ops.norm_project(obs, pred, U_obs, U_pred, None, None, 1, want_flag=False), 208),
        ("project obs only", lambda: ops.norm_project(obs, None, U_obs, None, None, None, 1, want_flag=False), 96),
        ("reconstruct S=1", lambda: ops.anchor_reconstruct(c_pred.view(6, n, 1), None, None, U_pred, None, 1, nrm=nrm), 136),
    ]:
        med, mn = timeit(fn)
        print(f"{name:22s} median {med*1e3:9.1f} us  min {mn*1e3:9.1f} us  {byt*n/med/1e6:8.1f} GB/s ({byt*n/med/1e6/8000*100:5.1f}% of 8 TB/s)")
    if n <= 2_000_000:
        S = 20
        cr = torch.randn(6, n, S, device=dev); A = torch.randn(6, S, device=dev)
        med, mn = timeit(lambda: ops.anchor_reconstruct(cr, A, None, U_pred, None, 1, nrm=nrm))
        print(f"{'reconstruct S=20':22s} median {med*1e3:9.1f} us  {2416*n/med/1e6:8.1f} GB/s")
if what in ("all", "km"):
    x = c_pred.contiguous()
    sh = ops.KMeansShard(x, 20)
    c0 = ops.kmeans_init_farthest(x, 20, 12345)
    med, mn = timeit(lambda: ops.kmeans_init_farthest(x, 20, 12345), reps=5, warm=1)
    print(f"{'kmeans init (19 steps)':22s} median {med*1e3:9.1f} us  {32*19*n/med/1e6:8.1f} GB/s")
    cen = c0.clone(); sh.scan(); sh.begin(n, cen)
    prev = None
    for it in range(int(sys.argv[3]) if len(sys.argv) > 3 else 24):
        a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        a.record(); part = sh.assign(cen); b.record()
        queued = None
        if os.environ.get("ET_FILTER_DEBUG") and it > 0:  # library built with -DET_FILTER_DEBUG: queued points in the NaN slot
            queued = int(part[-1]); part[-1] = 0
        sh.update(part, cen, 1e-4); c.record(); torch.cuda.synchronize()
        lab = sh.labels_u8[:n].clone()
        ch = float((lab != prev).float().mean()) if prev is not None else 1.0
        prev = lab
        print(f"iter {it:3d} assign+reduce {a.elapsed_time(b)*1e3:8.1f} us  update {b.elapsed_time(c)*1e3:6.1f} us  changed {ch*100:6.2f}%  -> {24*n/a.elapsed_time(b)/1e6:7.1f} GB/s" + (f"  queued {queued/n*100:.3f}%" if queued is not None else ""))
    med, mn = timeit(lambda: ops.kmeans_predict(x, cen), reps=5, warm=1)
    print(f"{'predict (sims only)':22s} median {med*1e3:9.1f} us  {24*n/med/1e6:8.1f} GB/s (+12 B/pt written)")
if what in ("all", "model"):
    # model form (S=20): forward, fused metrics, backward at N = 1e6
    m = min(n, 1_000_000)
    S = 20
    cr = torch.randn(6, m, S, device=dev); A = torch.randn(6, S, device=dev)
    nr = nrm[:, :m].contiguous(); gt = pred[:m].contiguous()
    med, mn = timeit(lambda: ops.anchor_reconstruct(cr, A, None, U_pred, None, 1, nrm=nr))
    print(f"{'reconstruct S=20 fwd':22s} median {med*1e3:9.1f} us  {2416*m/med/1e6:8.1f} GB/s ({2416*m/med/1e6/8000*100:5.1f}% of 8 TB/s)")
    med, mn = timeit(lambda: ops.anchor_reconstruct_metrics(cr, gt, A, None, U_pred, None, 1, nrm=nr))
    print(f"{'fused ADE/FDE S=20':22s} median {med*1e3:9.1f} us  {(480+16+96+8)*m/med/1e6:8.1f} GB/s (600 B/traj instead of 2416 + metrics pass)")
    from eigentrajectory_amd.ops import _reconstruct_bwd
    dt = torch.randn(S, m, 12, 2, device=dev)
    med, mn = timeit(lambda: _reconstruct_bwd(dt, None, nr, U_pred, None, 1, 0.0, 8))
    print(f"{'reconstruct S=20 bwd':22s} median {med*1e3:9.1f} us  {2416*m/med/1e6:8.1f} GB/s ({2416*m/med/1e6/8000*100:5.1f}% of 8 TB/s)")
    # scene-sized batches: launch-latency regime (ETH test scenes have <= 57 pedestrians)
    from eigentrajectory_amd import EigenTrajectory
    from eigentrajectory_amd.utils import DotDict, default_hyper_params
    class Zero(torch.nn.Module):
        def forward(self, x): return torch.zeros(6, x.size(1), 20, device=x.device)
    hooks = DotDict(model_forward_pre_hook=lambda c, o, a=None: torch.cat([c, o], 0), model_forward=lambda i, b: b(i), model_forward_post_hook=lambda o, a=None: o)
    model = EigenTrajectory(Zero(), hooks, default_hyper_params(static_dist=0.3)).to(dev)
    model.calculate_parameters(obs[:100000], pred[:100000])
    for nb in (5, 57, 128, 4096):
        o, p = obs[:nb].contiguous(), pred[:nb].contiguous()
        with torch.no_grad():
            for _ in range(5): model(o, p)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(200): out = model(o, p)
            torch.cuda.synchronize(); dt_f = (time.perf_counter() - t0) / 200
            t0 = time.perf_counter()
            for _ in range(200): a_, f_ = model.evaluate(o, p)
            torch.cuda.synchronize(); dt_e = (time.perf_counter() - t0) / 200
        print(f"wrapper forward (losses) N={nb:5d}: {dt_f*1e6:7.1f} us/scene   evaluate (fused ADE/FDE): {dt_e*1e6:7.1f} us/scene")
