#!/usr/bin/env python3
"""Same-process A/B of the fused metrics epilogue (S = 20, N = 1e7): vector-ALU kernel (ET_METRICS_MFMA=0) against the
matrix-core kernel with fp32 (f32) and two-term f16 (1, the default) matrix instructions; modes MOVING (the bench's) and SPLIT (what the wrapper's evaluate() runs)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigentrajectory_amd import ops  # noqa: E402
from eigentrajectory_amd import _lib as L  # noqa: E402
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


def med(fn, reps=10):
    for _ in range(2):
        fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


res = {}
for rnd in range(2):
    for mf in ("0", "f32", "1"):
        L.set_option("metrics_form", {"0": "t", "f32": "f", "1": "a"}[mf])
        t_mov = med(lambda: ops.anchor_reconstruct_metrics(C20, pred, A, None, Up, None, ops.MODE_MOVING, nrm=nrm))
        t_spl = med(lambda: ops.anchor_reconstruct_metrics(C20, pred, A, A, Up, Up, ops.MODE_SPLIT, 0.3, nrm=nrm))
        print(f"round {rnd} ET_METRICS_MFMA={mf}: MOVING {t_mov:.3f} ms ({600 * n / t_mov / 1e6 / 8000:.3f} of 8 TB/s)   "
              f"SPLIT {t_spl:.3f} ms ({600 * n / t_spl / 1e6 / 8000:.3f})", flush=True)
L.set_option("metrics_form", "t")
a0, f0 = ops.anchor_reconstruct_metrics(C20, pred, A, A, Up, Up, ops.MODE_SPLIT, 0.3, nrm=nrm)
for mf in ("f32", "1"):
    L.set_option("metrics_form", {"0": "t", "f32": "f", "1": "a"}[mf])
    a1, f1 = ops.anchor_reconstruct_metrics(C20, pred, A, A, Up, Up, ops.MODE_SPLIT, 0.3, nrm=nrm)
    print(mf, "max |ADE diff|", float((a0 - a1).abs().max()), "max |FDE diff|", float((f0 - f1).abs().max()),
          "of max ADE", float(a0.max()), flush=True)
