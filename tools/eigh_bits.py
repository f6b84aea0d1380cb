#!/usr/bin/env python3
"""Bit-level comparison of et_eigh_topk between two builds of the library (a restructured Jacobi kernel must give the
same U / sigma bits as the one before it):   python tools/eigh_bits.py dump out.npz   (with ET_LIBETAMD=<variant> or
not), then   python tools/eigh_bits.py cmp a.npz b.npz.  120 random problems: n = 1 ... 64, scales 1e-3 ... 1e3, zero rows,
repeated eigenvalues, plus the Gram matrices of the synthetic trajectories."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def problems():
    rng = np.random.default_rng(7)
    out = []
    for i in range(120):
        n = int(rng.integers(1, 65)) if i >= 8 else (1, 2, 3, 16, 24, 63, 64, 5)[i]
        x = rng.standard_normal((n, 3 * n + 2)) * 10.0 ** rng.uniform(-3, 3)
        if i % 7 == 3 and n > 2:
            x[rng.integers(0, n)] = 0.0
        if i % 11 == 5 and n > 3:
            x[1] = x[0]
        out.append((x @ x.T).astype(np.float64))
    return out


def main():
    if sys.argv[1] == "dump":
        import torch
        from eigentrajectory_amd import ops
        from eigentrajectory_amd.synth import synthetic_trajectories_torch
        dev = torch.device("cuda:0")
        mats = [torch.from_numpy(g).to(dev) for g in problems()]
        o, p = synthetic_trajectories_torch(200_000, dev, seed=0, min_disp=1e-3)
        g_obs, g_pred, _ = ops.fit_gram(o, p, ops.MODE_MOVING, 0.0, 1)
        g_obs, g_pred = torch.from_numpy(np.round(g_obs.cpu().numpy(), 3)).to(dev), torch.from_numpy(np.round(g_pred.cpu().numpy(), 3)).to(dev)
        mats += [g_obs, g_pred]  # (rounded: the Gram kernels of two builds may sum in different orders)
        res = {}
        for i, g in enumerate(mats):
            k = min(6, g.shape[0])
            U, s = ops.eigh_topk(g, k)
            res[f"U{i}"], res[f"s{i}"] = U.cpu().numpy(), s.cpu().numpy()
        (Uo, so), (Up, sp) = ops.eigh_topk_batch([g_obs, g_pred], 6)
        res["Ub0"], res["Ub1"] = Uo.cpu().numpy(), Up.cpu().numpy()
        np.savez(sys.argv[2], **res)
        print("dumped", len(res), "arrays ->", sys.argv[2])
    else:
        a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
        bad = [k for k in a.files if not np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32))]
        print(f"{len(a.files)} arrays, {len(bad)} differ" + (f": {bad[:10]}" if bad else " -- bit-identical"))
        for k in bad[:5]:
            print(k, np.abs(a[k] - b[k]).max())
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
