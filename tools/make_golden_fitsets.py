#!/usr/bin/env python3
"""Fit-set fixtures for ALL five ETH/UCY splits (train + val windows), through the reference's own loader.

    python tools/make_golden_fitsets.py --ref /root/reference --out tests/golden

A split's train/val directory holds the per-scene files of the OTHER scenes, and the same file
appears in up to four splits (8 unique train + 8 unique val files).  The reference's
TrajectoryDataset (utils/dataloader.py:154-241) windows every file independently and concatenates
the files in ``os.listdir`` order, so the fixture stores

* ``data/files/<file>.npz``  windows of ONE file (``q`` int32 = coordinates x 1e4, ``seq_start_end``),
  produced by running the reference's loader on a directory that holds only that file;
* ``data/splits.json``       for every (scene, phase) the file names in the order the reference's
  ``os.listdir`` returned them when the G2 fixtures (fitted U / anchors) were captured.

``tests/_golden.py:dataset`` re-assembles any split from these.  Only data is written; the reference
is imported, never copied.  The ETH split is also kept whole (``data/eth_{train,val}.npz``, written by
tools/make_golden.py); a CPU test checks that the re-assembly reproduces it bit for bit.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SCENES = ["eth", "hotel", "univ", "zara1", "zara2"]


def quantise(traj_f32):
    q = np.rint(traj_f32.astype(np.float64) * 1e4).astype(np.int32)
    back = (q.astype(np.float64) / 1e4).astype(np.float32)
    assert np.array_equal(back, traj_f32), "dataset is not 4-decimal exact"
    return q


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    out = os.path.join(os.path.abspath(args.out), "data")
    os.makedirs(os.path.join(out, "files"), exist_ok=True)
    sys.path.insert(0, args.ref)
    os.chdir(args.ref)
    from utils.dataloader import TrajectoryDataset

    splits = {}
    done = {}
    for scene in SCENES:
        for phase in ("train", "val"):
            d = os.path.join(args.ref, "datasets", scene, phase)
            names = [n for n in os.listdir(d)]  # the order the reference's loader sees (dataloader.py:177)
            splits[f"{scene}/{phase}"] = names
            for name in names:
                if name in done:
                    continue
                with tempfile.TemporaryDirectory() as tmp:
                    os.symlink(os.path.join(d, name), os.path.join(tmp, name))
                    ds = TrajectoryDataset(tmp + "/", obs_len=8, pred_len=12)
                full = torch.cat([ds.obs_traj, ds.pred_traj], dim=1).numpy()
                sse = np.asarray(ds.seq_start_end, dtype=np.int32)
                path = os.path.join(out, "files", name.replace(".txt", ".npz"))
                np.savez_compressed(path, q=quantise(full), seq_start_end=sse)
                done[name] = full.shape[0]
                print(f"  {name}: {full.shape[0]} peds, {len(sse)} scenes, {os.path.getsize(path) / 1024:.1f} KiB")
    with open(os.path.join(out, "splits.json"), "w") as f:
        json.dump(splits, f, indent=1, sort_keys=True)
    for key, names in splits.items():
        print(f"  {key}: {sum(done[n] for n in names)} peds")


if __name__ == "__main__":
    main()
