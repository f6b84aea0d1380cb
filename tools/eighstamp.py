#!/usr/bin/env python3
"""Where a Jacobi round of eigh_topk goes (library variant built with -DET_EXP_EIGHSTAMP:
tools/build_variant.sh eighstamp et_fit.hip "-DET_EXP_EIGHSTAMP"; run with ET_LIBETAMD pointing at it): s_memtime ticks
of the 24 x 24 problem's first wavefront by phase, for the Gram matrices of the bench's data."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigentrajectory_amd import _lib as L, ops  # noqa: E402
from eigentrajectory_amd.synth import synthetic_trajectories_torch  # noqa: E402

dev = torch.device("cuda:0")
obs, pred = synthetic_trajectories_torch(1_000_000, dev, seed=0, min_disp=1e-3)
g_obs, g_pred, _ = ops.fit_gram(obs, pred, ops.MODE_MOVING, 0.0, 1)
fn = L.lib().et_debug_eighstamp
buf = (C.c_ulonglong * 8)()
ops.eigh_topk_batch([g_obs, g_pred], 6)
assert fn(buf, 1) == 0
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
ops.eigh_topk_batch([g_obs, g_pred], 6)
b.record()
torch.cuda.synchronize()
assert fn(buf, 0) == 0
v = [int(x) for x in buf]
rounds, sweeps = v[0], v[6]
tot = sum(v[1:6])
print(f"{a.elapsed_time(b) * 1e3:.1f} us between events (with the stamps), {sweeps} sweep checks, {rounds} rounds, {tot} ticks "
      f"({tot / max(rounds, 1):.0f} per round)")
for i, nm in enumerate(["block / V items of the wavefront", "look-ahead entries (+ shuffles)", "rotation chain + stepping",
                        "barrier", "convergence checks (per sweep)"]):
    print(f"    {nm:34s} {v[i + 1]:9d} ticks  {100 * v[i + 1] / tot:5.1f} %   {v[i + 1] / max(rounds, 1):6.0f} per round")
