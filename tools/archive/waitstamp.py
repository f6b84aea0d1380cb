#!/usr/bin/env python3
"""Where a pass of the packed Lloyd body spends its time (library variant built with -DET_EXP_WAITSTAMP:
tools/build_variant.sh waitstamp et_kmeans.hip "-DET_EXP_WAITSTAMP"; run with ET_LIBETAMD pointing at it).
One k-means fit at N (default 1e7), 100 iterations; prints cycles per pass: exposed load wait / whole pass / drains, and
the phases of a launch (prologue loads, fold, update, tables, passes, tail) as thread 0 of a workgroup sees them."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigentrajectory_amd import _lib as L, ops  # noqa: E402
from eigentrajectory_amd.synth import synthetic_trajectories_torch  # noqa: E402

dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
obs, pred = synthetic_trajectories_torch(n, dev, seed=0, min_disp=1e-3)
g_obs, g_pred, _ = ops.fit_gram(obs, pred, ops.MODE_MOVING, 0.0, 1)
(Uo, _), (Up, _) = ops.eigh_topk_batch([g_obs, g_pred], 6)
_, c, _, _ = ops.norm_project(obs, pred, Uo, Up, None, None, ops.MODE_MOVING, want_flag=False, want_nrm=False)
del obs, pred
c0 = ops.kmeans_init_farthest(c, 20, 12345)
ops.kmeans_fit(c, c0, 5, 1e-4, trace=False)
fn = L.lib().et_debug_waitstamp
buf = (C.c_ulonglong * 8)()
assert fn(buf, 1) == 0
res = ops.kmeans_fit(c, c0, 100, 1e-4, timing=True, trace=False)
assert fn(buf, 0) == 0
passes, wait, total, drain, nd, _, waves = [int(v) for v in buf[:7]]
print(f"N = {n}: {res['n_iter']} iterations, avg launch {res['assign_ms'] / max(res['assign_launches'], 1) * 1e3:.1f} us "
      f"(with the stamps' own overhead)")
print(f"passes {passes} over {waves} wavefront-launches ({passes / max(waves, 1):.1f} per wavefront and launch)")
passes = max(passes, 1)  # (shards below the packed path's threshold: no packed passes, only the launch phases below)
total = max(total, 1)
print(f"per pass: whole {total / passes:.0f} cycles, exposed load wait {wait / passes:.0f} ({100 * wait / total:.1f} %), "
      f"queue drains {drain / passes:.0f} ({100 * drain / total:.1f} %; {nd} drains, {drain / max(nd, 1):.0f} cycles each)")
# where a launch goes (thread 0 of every workgroup)
fn2 = getattr(L.lib(), "et_debug_prostamp", None)
if fn2 is not None:
    buf2 = (C.c_ulonglong * 16)()
    assert fn2(buf2, 1) == 0
    ops.kmeans_fit(c, c0, 100, 1e-4, trace=False)
    assert fn2(buf2, 0) == 0
    v = [int(x) for x in buf2]
    nwg = max(v[0], 1)
    names = ["entry -> prologue loads arrived", "fold + barrier", "update (new centroids, error, flags)", "barrier + publish",
             "tables, matrix operand, barrier", "the pass loop", "final drain + barrier", "copies -> one, barrier, emit"]
    tot = sum(v[1:9])
    print(f"a launch as thread 0 of a workgroup sees it ({nwg} workgroup-launches; the last three rows carry this build's own "
          f"per-wavefront atomics):")
    for i, nm in enumerate(names):
        print(f"    {nm:40s} {v[i + 1] / nwg:8.0f} cycles  {v[i + 1] / nwg / 2.4e3:6.2f} us")
