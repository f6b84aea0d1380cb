#!/usr/bin/env python3
"""HBM ceilings of the box this runs on, with stock PyTorch-ROCm kernels on a 4.8 GB fp32 buffer (far beyond the 256 MB
Infinity Cache): pure write (fill), pure read (sum), copy.  The streaming kernels of libetamd.so are judged against
the 8 TB/s spec peak in bench.py; these numbers say how much of the gap is the memory system itself.
usage: python tools/bw_ceiling.py  (on the GPU box)"""
import json

import numpy as np
import torch


def med_ms(fn, reps=7):
    fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def main():
    dev = torch.device("cuda:0")
    n = 1_200_000_000
    x = torch.empty((n,), device=dev)
    y = torch.empty((n,), device=dev)
    gb = 4.0 * n / 1e9
    out = {"buffer_GB": gb}
    out["fill_write_GBs"] = round(gb / med_ms(lambda: x.fill_(1.0)) * 1e3, 1)
    out["sum_read_GBs"] = round(gb / med_ms(lambda: x.sum()) * 1e3, 1)
    out["copy_read_plus_write_GBs"] = round(2 * gb / med_ms(lambda: y.copy_(x)) * 1e3, 1)
    out["add_2read_1write_GBs"] = round(3 * gb / med_ms(lambda: torch.add(x, y, out=y)) * 1e3, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
