#!/usr/bin/env python3
"""Where a pass of the fused metrics kernel spends its time (library variant built with -DET_EXP_METSTAMP:
tools/build_variant.sh metstamp et_descriptor.hip "-DET_EXP_METSTAMP"; run with ET_LIBETAMD pointing at it).
S = 20, N = 1e7 (or argv[1]), modes MOVING and SPLIT; prints shader cycles per pass and wavefront by phase."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigentrajectory_amd import _lib as L, ops  # noqa: E402
from eigentrajectory_amd.synth import synthetic_trajectories_torch  # noqa: E402

dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
S = 20
obs, pred = synthetic_trajectories_torch(n, dev, seed=0, min_disp=1e-3)
g_obs, g_pred, _ = ops.fit_gram(obs, pred, ops.MODE_MOVING, 0.0, 1)
(Uo, _), (Up, _) = ops.eigh_topk_batch([g_obs, g_pred], 6)
_, _, nrm, _ = ops.norm_project(obs, pred, Uo, Up, None, None, ops.MODE_MOVING, want_flag=False)
C20 = torch.randn((6, n, S), device=dev) * 0.1
A = torch.randn((6, S), device=dev)
fn = L.lib().et_debug_metstamp
buf = (C.c_ulonglong * 8)()
names = ["hand-over + stores + requests", "wait for the pass's inputs", "LDS reads, gt normalisation, operands",
         "matrix instructions + hand-over", "distances", "best of S"]
for label, call in (("MOVING", lambda: ops.anchor_reconstruct_metrics(C20, pred, A, None, Up, None, ops.MODE_MOVING, nrm=nrm)),
                    ("SPLIT", lambda: ops.anchor_reconstruct_metrics(C20, pred, A, A, Up, Up, ops.MODE_SPLIT, 0.3, nrm=nrm))):
    call()
    assert fn(buf, 1) == 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    call()
    b.record()
    torch.cuda.synchronize()
    assert fn(buf, 0) == 0
    v = [int(x) for x in buf]
    passes, waves = v[0], v[7]
    tot = sum(v[1:7])
    print(f"{label}: {a.elapsed_time(b):.3f} ms (with the stamps' own overhead), {waves} wavefronts, {passes / waves:.0f} passes each, "
          f"{tot / passes:.0f} cycles per pass and wavefront")
    for i, nm in enumerate(names):
        print(f"    {nm:40s} {v[i + 1] / passes:7.0f} cycles  {100 * v[i + 1] / tot:5.1f} %")
