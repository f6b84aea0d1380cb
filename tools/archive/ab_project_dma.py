"""HISTORICAL (round 5): needs the `project_form` option of the commit before profiles/r05d_project_dma_ab.txt (the LDS-DMA
variant left the library: tools/lost_forms/project_lds_dma.hip.txt).

Same-process A/B of the projection kernel's row staging at N = 1e7 (verdict r4 item 2): rows through registers (default)
against memory -> LDS directly (buffer_load_dwordx4 ... lds), default and non-temporal policy.  HIP-event medians of 30
launches, alternating rounds; outputs compared bit for bit.  python tools/ab_project_dma.py [N]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from eigentrajectory_amd import ops, _lib as L
from eigentrajectory_amd.synth import synthetic_trajectories_torch

dev = torch.device("cuda:0")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
o, p = synthetic_trajectories_torch(n, dev, seed=0, min_disp=1e-3)
g_obs, g_pred, _ = ops.fit_gram(o, p, ops.MODE_MOVING, 0.0, 1)
(U_obs, _), (U_pred, _) = ops.eigh_topk_batch([g_obs, g_pred], 6)
ref = None
for rnd in range(3):
    for form in ("a", "d", "n"):
        L.set_option("project_form", form)
        out = ops.norm_project(o, p, U_obs, U_pred, None, None, ops.MODE_MOVING, want_flag=False)
        if ref is None:
            ref = [t.clone() for t in out[:3]]
        same = all(torch.equal(a, b) for a, b in zip(out[:3], ref))
        ts = []
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.norm_project(o, p, U_obs, U_pred, None, None, ops.MODE_MOVING, want_flag=False)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = float(np.median(ts))
        print(f"round {rnd} project_form={form}: {t:.4f} ms  ({208 * n / t / 1e6 / 8000:.4f} of 8 TB/s algorithmic, {224 * n / t / 1e9:.2f} TB/s moved)  same bits: {same}", flush=True)
L.set_option("project_form", "a")
