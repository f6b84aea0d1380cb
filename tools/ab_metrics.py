#!/usr/bin/env python3
"""Same-box A/B of the fused best-of-S metrics kernel (S = 20, N = 1e7, MOVING and SPLIT) between library variants
(tools/build_variant.sh): one subprocess per (round, library), alternating.   python tools/ab_metrics.py base <name> ... [rounds]"""
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from eigentrajectory_amd import ops
from eigentrajectory_amd.synth import synthetic_trajectories_torch
dev = torch.device("cuda:0")
n, S = 10_000_000, 20
obs, pred = synthetic_trajectories_torch(n, dev, seed=0, min_disp=1e-3)
g_obs, g_pred, _ = ops.fit_gram(obs, pred, ops.MODE_MOVING, 0.0, 1)
(Uo, _), (Up, _) = ops.eigh_topk_batch([g_obs, g_pred], 6)
_, _, nrm, _, pose = ops.norm_project(obs, pred, Uo, Up, None, None, ops.MODE_MOVING, want_flag=False, want_pose=True)
_, _, _, _, pose_s = ops.norm_project(obs, pred, Uo, Up, Uo, Up, ops.MODE_SPLIT, 0.3, want_flag=False, want_pose=True)
C20 = torch.randn((6, n, S), device=dev) * 0.1
A = torch.randn((6, S), device=dev)
def med(fn, reps=12):
    for _ in range(3): fn()
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
from eigentrajectory_amd import _lib as L
import os
if os.environ.get("ET_AB_FORM"): L.set_option("metrics_form", os.environ["ET_AB_FORM"])
t_mov = med(lambda: ops.anchor_reconstruct_metrics(C20, pred, A, None, Up, None, ops.MODE_MOVING, nrm=nrm))
t_spl = med(lambda: ops.anchor_reconstruct_metrics(C20, pred, A, A, Up, Up, ops.MODE_SPLIT, 0.3, nrm=nrm))
p_mov = med(lambda: ops.anchor_reconstruct_metrics(C20, pred, A, None, Up, None, ops.MODE_MOVING, pose=pose))
p_spl = med(lambda: ops.anchor_reconstruct_metrics(C20, pred, A, A, Up, Up, ops.MODE_SPLIT, 0.3, pose=pose_s))
t_prj = med(lambda: ops.norm_project(obs, pred, Uo, Up, None, None, ops.MODE_MOVING, want_flag=False))
p_prj = med(lambda: ops.norm_project(obs, pred, Uo, Up, None, None, ops.MODE_MOVING, want_flag=False, want_pose=True))
print("MOVING %%.4f ms (%%.3f of 8 TB/s)  SPLIT %%.4f ms (%%.3f) | with pose: MOVING %%.4f (%%.3f)  SPLIT %%.4f (%%.3f) | project %%.4f, + pose %%.4f" %% (t_mov, 600 * n / t_mov / 1e6 / 8000, t_spl, 600 * n / t_spl / 1e6 / 8000, p_mov, 600 * n / p_mov / 1e6 / 8000, p_spl, 600 * n / p_spl / 1e6 / 8000, t_prj, p_prj))
''' % R
names = [a for a in sys.argv[1:] if not a.isdigit()]
rounds = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 3
for r in range(rounds):
    for name in names:
        env = dict(os.environ)
        if name.startswith("form="):  # the shipped library under option metrics_form = <letter>
            env["ET_AB_FORM"] = name[5:]
        elif name != "base":
            env["ET_LIBETAMD"] = os.path.join(R, "eigentrajectory_amd", "variants", f"libetamd_{name}.so")
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.strip().splitlines() if l.startswith("MOVING")]
        print(f"round {r} {name:10s} " + (line[-1] if line else "FAILED: " + out.stderr[-400:]), flush=True)
