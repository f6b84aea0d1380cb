"""Same-box A/B of the S = 20 reconstruction kernels (run with ET_LIBETAMD pointing at a variant build)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from eigentrajectory_amd import ops
dev = torch.device("cuda:0")
n, S = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000, 20
g = torch.Generator(device=dev).manual_seed(1)
C = torch.randn((6, n, S), device=dev, generator=g) * 0.1
A = torch.randn((6, S), device=dev, generator=g)
U = torch.linalg.qr(torch.randn((24, 6), device=dev, generator=g))[0].contiguous()
nrm = torch.randn((4, n), device=dev, generator=g)
gt = torch.randn((n, 12, 2), device=dev, generator=g)
def t(fn, reps=7):
    fn(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
with torch.no_grad():
    rec = ops.anchor_reconstruct(C, A, None, U, None, 1, nrm=nrm)
    chk = float(rec.double().abs().sum())
    chk_b = float(ops._reconstruct_bwd(rec, None, nrm, U, None, 1, 0.0, 8).double().abs().sum())
    f = t(lambda: ops.anchor_reconstruct(C, A, None, U, None, 1, nrm=nrm))
    b = t(lambda: ops._reconstruct_bwd(rec, None, nrm, U, None, 1, 0.0, 8))
    m = t(lambda: ops.anchor_reconstruct_metrics(C, gt, A, None, U, None, 1, nrm=nrm))
print(os.path.basename(os.environ.get("ET_LIBETAMD", "default")), f"checksum {chk:.6e} bwd {chk_b:.9e}",
      f"fwd {f:.3f} ms {2416*n/f/1e6:.0f} GB/s | bwd {b:.3f} ms {2416*n/b/1e6:.0f} GB/s | metrics {m:.3f} ms {600*n/m/1e6:.0f} GB/s")
