#!/usr/bin/env python3
"""Randomized soak of the fused metrics epilogue (ON THE GPU BOX): the persistent matrix-core kernel (LDS-DMA ring, counted
waits, f16 split / fp32 instructions) against the vector-ALU tile kernel (ET_METRICS_MFMA=0) on random shapes, modes,
normaliser sources and value ranges.    python tools/soak_metrics.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigentrajectory_amd import ops  # noqa: E402
from eigentrajectory_amd import _lib as L  # noqa: E402
from eigentrajectory_amd.synth import synthetic_trajectories_torch  # noqa: E402

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
bad = 0
for it in range(cases):
    S = int(rng.integers(12, 65))
    n = int(rng.choice([rng.integers(64 // S, 40), rng.integers(40, 3000), rng.integers(3000, 400000)]))
    mode = int(rng.integers(0, 4))
    obs, gt = synthetic_trajectories_torch(n, dev, seed=int(rng.integers(1 << 30)), min_disp=1e-3 if mode == 1 else 0.0)
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
    um, us_ = (torch.randn((24, 6), device=dev, generator=g) * 0.3 for _ in range(2))
    a_m, a_s = (torch.randn((6, S), device=dev, generator=g) for _ in range(2))
    scale = float(rng.choice([0.1, 1.0, 1.0, 30.0, 1000.0]))  # 1000: leaves the f16 range -> fp32 instructions
    c = torch.randn((6, n, S), device=dev, generator=g) * scale
    if rng.random() < 0.2:
        c[int(rng.integers(6)), int(rng.integers(n)), int(rng.integers(S))] = float("nan")
    kw = {}
    src = rng.random()
    if mode != ops.MODE_IDENTITY and src < 0.66:  # cached normaliser state instead of the observed rows: nrm, or the pose record
        z = torch.zeros((8 * 2, 6), device=dev)
        _, _, nrm, _, pose = ops.norm_project(obs, None, z, None, z, None, mode, 0.3, want_flag=False, want_pose=True)
        kw["nrm"] = nrm  # (the tile kernel reads nrm; the matrix-core kernel the pose when it is given)
        if src < 0.33:
            kw["pose"] = pose
    else:
        kw["obs"] = obs
    out = {}
    for v in ("0", "1"):
        L.set_option("metrics_form", {"0": "t", "f32": "f", "1": "a"}.get(v, v))
        out[v] = ops.anchor_reconstruct_metrics(c, gt, a_m, a_s, um, us_, mode, 0.3, **kw)
    for name, a, b in (("ade", out["0"][0], out["1"][0]), ("fde", out["0"][1], out["1"][1])):
        na, nb = torch.isnan(a), torch.isnan(b)
        if not torch.equal(na, nb):
            bad += 1
            print(f"case {it}: NaN pattern differs ({name}) n={n} S={S} mode={mode} scale={scale} {list(kw)}", flush=True)
            continue
        fin = ~na
        if fin.any():
            ref = float(a[fin].abs().max())
            err = float((a[fin] - b[fin]).abs().max()) / max(ref, 1e-30)
            worst = max(worst, err)
            if err > 3e-6:
                bad += 1
                print(f"case {it}: {name} differs by {err:.2e} of the largest value  n={n} S={S} mode={mode} scale={scale} {list(kw)}", flush=True)
torch.cuda.synchronize()
print(f"{cases} cases, {bad} mismatches, worst |difference| / largest value {worst:.2e}")
sys.exit(1 if bad else 0)
