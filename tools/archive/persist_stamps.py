#!/usr/bin/env python3
"""Phase timeline of the persistent Lloyd kernel (development aid).  Needs a library built with -DET_PERSIST_STAMPS:
    tools/build_variant.sh stamps et_kmeans.hip -DET_PERSIST_STAMPS
    ET_LIBETAMD=$PWD/eigentrajectory_amd/variants/libetamd_stamps.so python tools/persist_stamps.py [N]
Prints, per phase, the median over iterations 20..90 of: the spread of the workgroups' arrivals, last arrival -> last go,
fold, update, body (min / median / max over workgroups), all in microseconds."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigentrajectory_amd import ops, _lib
from eigentrajectory_amd.synth import synthetic_trajectories_torch

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0")
obs, pred = synthetic_trajectories_torch(n, dev, seed=0, min_disp=1e-3)
g_obs, g_pred, _ = ops.fit_gram(obs, pred, 1, 0.0, 1)
(U_obs, _), (U_pred, _) = ops.eigh_topk_batch([g_obs, g_pred], 6)
_, c_pred, _, _ = ops.norm_project(obs, pred, U_obs, U_pred, None, None, 1, want_flag=False)
x = c_pred.contiguous()
c0 = ops.kmeans_init_farthest(x, 20, 12345 % n)
for _ in range(2):
    res = ops.kmeans_fit(x, c0, 100, 1e-4, trace=False, timing=True)
print("fit: %d iterations, launch %.3f ms" % (res["n_iter"], res["assign_ms"]))
W, I, Kk = 256, 104, 10
buf = np.zeros((W, I, Kk), dtype=np.uint64)
rc = _lib.lib().et_debug_persist_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes))
assert rc == 0
t = buf.astype(np.int64)
used = (t[:, 50, 0] != 0)
t = t[used]
print("workgroups:", t.shape[0])
rows = []
for it in range(20, 90):
    top, go, folded, updated, b0, b1 = (t[:, it, k] for k in range(6))
    s6, s7, s8, s9 = (t[:, it, k] for k in range(6, 10))
    nxt = t[:, it + 1, 0]
    rows.append([
        (top.max() - top.min()), (go.max() - top.max()), (go.min() - top.max()),
        np.median(folded - go), np.median(updated - folded), np.median(b0 - updated),
        (b1 - b0).min(), np.median(b1 - b0), (b1 - b0).max(), np.median(nxt - b1), (nxt.max() - top.max()),
        np.median(s6 - b0), np.median(s7 - s6), np.median(s8 - s7), np.median(s9 - s8),
    ])
r = np.median(np.array(rows, dtype=np.float64), axis=0) * 0.01
names = ["arrival spread", "last arrival -> last go", "last arrival -> first go", "fold", "update", "update -> body", "body min",
         "body median", "body max", "body end -> arrival", "iteration (last arrival to last arrival)",
         "  body: start -> operands staged", "  body: passes (wavefront 0)", "  body: final queue drain (wavefront 0)",
         "  body: barrier + emit"]
for nm, v in zip(names, r):
    print(f"{nm:42s} {v:8.2f} us")
# where does the spread of the body times come from?  per workgroup (mean over iterations), per XCD (workgroup id mod 8)
body = (t[:, 20:90, 5] - t[:, 20:90, 4]).astype(np.float64) * 0.01
pw = body.mean(axis=1)
print("per-workgroup mean body time: min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f us" %
      (pw.min(), np.percentile(pw, 10), np.median(pw), np.percentile(pw, 90), pw.max()))
print("within-workgroup std over iterations (median over workgroups): %.2f us" % np.median(body.std(axis=1)))
if t.shape[0] >= 8:
    print("by workgroup id mod 8:", " ".join("%.2f" % pw[k::8].mean() for k in range(8)))
    print("by workgroup id // 32:", " ".join("%.2f" % pw[k * 32:(k + 1) * 32].mean() for k in range(t.shape[0] // 32)))
order = np.argsort(pw)
print("slowest workgroups:", order[-12:].tolist(), "fastest:", order[:12].tolist())
# per iteration: max over workgroups minus the median
print("per-iteration (max - median) body: median %.2f us" % np.median(body.max(axis=0) - np.median(body, axis=0)))
