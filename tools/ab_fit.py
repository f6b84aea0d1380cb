#!/usr/bin/env python3
"""Same-box A/B of the fit stage between library variants (tools/build_variant.sh): Gram launch sequence and eigensolve
timed separately with HIP events at N = 1e7 (median of 20 lone calls), alternating the libraries over `rounds`.
    python tools/ab_fit.py base head gramnt [rounds]          (worker: python tools/ab_fit.py --worker)"""
import json
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)


def worker():
    import numpy as np
    import torch
    from eigentrajectory_amd import _lib as L, ops
    from eigentrajectory_amd.synth import synthetic_trajectories_torch
    dev = torch.device("cuda:0")
    n = int(float(os.environ.get("AB_N", "1e7")))
    obs, pred = synthetic_trajectories_torch(n, dev, seed=0, min_disp=1e-3)

    def med(fn, reps=20):
        for _ in range(3):
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

    g_obs, g_pred, _ = ops.fit_gram(obs, pred, ops.MODE_MOVING, 0.0, 1)
    out = dict(gram_ms=med(lambda: ops.fit_gram(obs, pred, ops.MODE_MOVING, 0.0, 1)),
               eigh_ms=med(lambda: ops.eigh_topk_batch([g_obs, g_pred], 6)),
               fit_ms=med(lambda: ops.eigh_topk_batch(list(ops.fit_gram(obs, pred, ops.MODE_MOVING, 0.0, 1)[:2]), 6)),
               fit_one_call_ms=(med(lambda: ops.fit_descriptor(obs, pred, 6, ops.MODE_MOVING, 0.0, 1)) if hasattr(ops, "fit_descriptor")
                                and hasattr(L.lib(), "et_fit_descriptor") else float("nan")),
               g00=float(g_obs[0, 0].item()))
    print(json.dumps(out))


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
                print(f"round {r} {name:8s} gram {d['gram_ms']:.4f}  eigh {d['eigh_ms']:.4f}  fit {d['fit_ms']:.4f}  one call {d['fit_one_call_ms']:.4f}  (G_obs[0,0] = {d['g00']!r})", flush=True)
            except Exception:
                print(f"round {r} {name}: FAILED\n{res.stdout[-500:]}\n{res.stderr[-1500:]}", flush=True)


if __name__ == "__main__":
    worker() if "--worker" in sys.argv else main()
