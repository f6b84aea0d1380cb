"""Where a launch of the fast reference-order Lloyd kernel goes (build: tools/build_variant.sh rfstamp et_kmeans_reforder.hip
-DET_EXP_RFSTAMP; run with ET_LIBETAMD=eigentrajectory_amd/variants/libetamd_rfstamp.so).  Stamps (100 MHz clock) of workgroup 0,
the tail's workgroup and the last workgroup of the LAST launch of a fit: python tools/rfstamp.py [sizes...]"""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from eigentrajectory_amd import ops, _lib as L
from eigentrajectory_amd.synth import gaussian_points_np

dev = torch.device("cuda:0")
names = {0: ["start", "prologue", "assign", "levels 0-1 + store", "counts"],
         1: ["start", "prologue", "assign", "levels 0-1 + store", "counts"],
         2: ["start", "level 2 + store", "arrival", "-", "-"],
         3: ["start", "level 3 + centroids", "batch arrival", "error", "state + counters"]}
for n in [int(float(a)) for a in sys.argv[1:]] or [100000, 1000000, 10000000]:
    x = torch.from_numpy(gaussian_points_np(6, n, seed=3, n_blobs=7)).to(dev)
    c0 = ops.kmeans_init_farthest(x, 20, 17)
    buf0 = (C.c_ulonglong * 64)()
    L.lib().et_debug_rfstamps(buf0)  # (reading resets the accumulated slots)
    ops.kmeans_fit_reference_order(x, c0, 11, -1.0, trace=False)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    assert L.lib().et_debug_rfstamps(buf) == 0
    st = np.array(buf[:], dtype=np.int64).reshape(4, 16)
    t0 = min(st[0, 0], st[1, 0])
    print(f"N = {n}")
    for who, label, cnt in ((0, "groups kernel, workgroup 0", 5), (1, "groups kernel, tail", 5), (2, "update kernel, workgroup 0", 3),
                            (3, "update kernel, last", 5)):
        row = st[who, :cnt]
        print(f"  {label:27s} starts at {(row[0] - t0) / 100:7.2f} us; " +
              ", ".join(f"{names[who][i]} {(row[i] - row[i - 1]) / 100:.2f}" for i in range(1, cnt)) +
              f"; ends at {(row[cnt - 1] - t0) / 100:.2f} us")
    acc = st[3, 8:14] / 100.0 / 11  # the warm-up call in front + ... : per launch of the 11 launches since the last read
    print("  workgroup 0, levels 0-1 per launch: zeroing %.2f, loads %.2f, updates %.2f, barrier %.2f, level 1 %.2f, barrier %.2f us" % tuple(acc))
    print(f"  groups kernel, last workgroup to pass: assign {(st[0, 8] - t0) / 100:.2f}, levels {(st[0, 9] - t0) / 100:.2f}, end {(st[0, 10] - t0) / 100:.2f} us")
