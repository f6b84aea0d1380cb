#!/usr/bin/env python3
"""Same-box A/B of the reconstruction between library variants (tools/build_variant.sh): S = 1 and S = 20 forward at N = 1e7
between HIP events (median of 10 lone calls), alternating the libraries.   python tools/ab_recon.py base nostream [rounds]"""
import json
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)


def worker():
    import numpy as np
    import torch
    from eigentrajectory_amd import ops
    from eigentrajectory_amd.synth import synthetic_trajectories_torch
    dev = torch.device("cuda:0")
    n = 10_000_000
    obs, pred = synthetic_trajectories_torch(n, dev, seed=0, min_disp=1e-3)
    U_obs, U_pred = ops.fit_descriptor(obs, pred, 6, ops.MODE_MOVING, 0.0, 1)[:2]
    _, c_pred, nrm, _ = ops.norm_project(obs, pred, U_obs, U_pred, None, None, ops.MODE_MOVING, want_flag=False)
    del obs, pred
    C20 = torch.randn((6, n, 20), device=dev) * 0.1
    A = torch.randn((6, 20), device=dev)

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

    print(json.dumps(dict(s1=med(lambda: ops.anchor_reconstruct(c_pred.view(6, n, 1), None, None, U_pred, None, ops.MODE_MOVING, nrm=nrm)),
                          s20=med(lambda: ops.anchor_reconstruct(C20, A, None, U_pred, None, ops.MODE_MOVING, nrm=nrm)))))


def main():
    names = [a for a in sys.argv[1:] if not a.isdigit()]
    rounds = int(next((a for a in sys.argv[1:] if a.isdigit()), "3"))
    for r in range(rounds):
        for name in names:
            env = dict(os.environ)
            if name != "base":
                env["ET_LIBETAMD"] = os.path.join(R, "eigentrajectory_amd", "variants", f"libetamd_{name}.so")
            res = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], env=env, capture_output=True, text=True)
            try:
                d = json.loads(res.stdout.strip().splitlines()[-1])
                print(f"round {r} {name:9s} S=1 {d['s1']:.4f} ms ({136e7 / d['s1'] / 1e6 / 8000:.3f})   S=20 {d['s20']:.4f} ms ({2416e7 / d['s20'] / 1e6 / 8000:.3f})", flush=True)
            except Exception:
                print(f"round {r} {name}: FAILED\n{res.stdout[-300:]}\n{res.stderr[-1200:]}", flush=True)


if __name__ == "__main__":
    worker() if "--worker" in sys.argv else main()
