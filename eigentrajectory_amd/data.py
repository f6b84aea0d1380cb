"""ETH/UCY ingestion for the descriptor path: text files -> (N, T, 2) windows.

Restates what the reference's loader produces (utils/dataloader.py:154-241) so that the fit and the
ADE/FDE parity configs can run from the raw `frame<TAB>ped<TAB>x<TAB>y` files:

* a window is `obs_len + pred_len` CONSECUTIVE entries of the file's sorted unique frame ids,
  starting at every frame (skip = 1);
* a pedestrian belongs to a window only if it is present from the window's first to its last frame
  (dataloader.py:204-207); coordinates are rounded to 4 decimals (:202);
* a window becomes a scene only if MORE THAN ONE pedestrian qualifies (`min_ped=1` with a strict
  `>`, dataloader.py:160,216);
* scenes are concatenated file by file; `seq_start_end` delimits them.

Not restated: the `non_linear_ped` poly-fit flag and the all-ones `loss_mask` (unused by the
descriptor path).  File order: the reference uses `os.listdir` (arbitrary); here files are sorted
by name unless an explicit list is given.  Host-side numpy; plain or gzip-compressed text.
"""
from __future__ import annotations

import gzip
import os

import numpy as np
import torch


def read_track_file(path, delim="\t"):
    """-> float64 array (rows, 4): frame, ped, x, y"""
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt") as f:
        rows = [[float(v) for v in line.strip().split(delim)] for line in f if line.strip()]
    return np.asarray(rows, dtype=np.float64)


def windows_from_tracks(data, obs_len=8, pred_len=12, skip=1, min_ped=1):
    """All scenes of one file.  Returns (list of (n_i, seq_len, 2) float64 arrays)."""
    seq_len = obs_len + pred_len
    frames = np.unique(data[:, 0])
    frame_index = np.searchsorted(frames, data[:, 0])          # row -> position of its frame
    order = np.argsort(frame_index, kind="stable")              # rows grouped by frame, file order kept
    data, frame_index = data[order], frame_index[order]
    first_row = np.searchsorted(frame_index, np.arange(len(frames) + 1))
    scenes = []
    for start in range(0, len(frames) - seq_len + 1, skip):
        rows = data[first_row[start]:first_row[start + seq_len]]
        fidx = frame_index[first_row[start]:first_row[start + seq_len]] - start
        peds = np.unique(rows[:, 1])
        keep = []
        for ped in peds:
            sel = rows[:, 1] == ped
            f = fidx[sel]
            if f[0] != 0 or f[-1] != seq_len - 1:
                continue  # not present over the whole window (dataloader.py:204-207)
            xy = np.around(rows[sel, 2:4], decimals=4)
            if xy.shape[0] != seq_len:
                raise ValueError("pedestrian with a gap inside a window: the reference loader fails here too")
            keep.append(xy)
        if len(keep) > min_ped:
            scenes.append(np.stack(keep, axis=0))
    return scenes


class TrajectoryData:
    """obs_traj (N,obs_len,2), pred_traj (N,pred_len,2) float32 tensors + scene bookkeeping
    (the attributes utils/trainer.py:51-52 reads from the reference's dataset object)."""

    def __init__(self, data_dir, obs_len=8, pred_len=12, skip=1, min_ped=1, delim="\t", files=None):
        self.obs_len, self.pred_len = obs_len, pred_len
        if files is None:
            files = sorted(f for f in os.listdir(data_dir) if f.endswith((".txt", ".txt.gz")))
        scenes = []
        for name in files:
            scenes += windows_from_tracks(read_track_file(os.path.join(data_dir, name), delim), obs_len, pred_len, skip,
                                          min_ped)
        self.num_seq = len(scenes)
        self.num_peds_in_seq = np.array([s.shape[0] for s in scenes], dtype=np.int64)
        full = np.concatenate(scenes, axis=0) if scenes else np.zeros((0, obs_len + pred_len, 2))
        self.obs_traj = torch.from_numpy(full[:, :obs_len]).type(torch.float).contiguous()
        self.pred_traj = torch.from_numpy(full[:, obs_len:]).type(torch.float).contiguous()
        ends = np.cumsum(self.num_peds_in_seq)
        self.seq_start_end = [(int(e - n), int(e)) for e, n in zip(ends, self.num_peds_in_seq)]

    @classmethod
    def from_arrays(cls, obs, pred, seq_start_end):
        """The same object from windows that are already assembled: obs (N,obs_len,2), pred (N,pred_len,2) and the scenes'
        (start, end) row ranges -- preprocessed splits, synthetic data."""
        self = cls.__new__(cls)
        self.obs_traj = torch.as_tensor(obs).type(torch.float).contiguous()
        self.pred_traj = torch.as_tensor(pred).type(torch.float).contiguous()
        self.obs_len, self.pred_len = int(self.obs_traj.shape[1]), int(self.pred_traj.shape[1])
        self.seq_start_end = [(int(s), int(e)) for s, e in seq_start_end]
        self.num_seq = len(self.seq_start_end)
        self.num_peds_in_seq = np.array([e - s for s, e in self.seq_start_end], dtype=np.int64)
        return self

    def __len__(self):
        return self.num_seq

    def __getitem__(self, index):
        s, e = self.seq_start_end[index]
        return self.obs_traj[s:e], self.pred_traj[s:e]


def scene_batches(num_peds_in_seq, batch_size, shuffle=False, drop_last=False, generator=None):
    """Scene indices grouped until a batch holds at least `batch_size` pedestrians
    (TrajBatchSampler, utils/dataloader.py:68-119)."""
    n = len(num_peds_in_seq)
    order = torch.randperm(n, generator=generator).tolist() if shuffle else list(range(n))
    batch, total = [], 0
    for idx in order:
        batch.append(idx)
        total += int(num_peds_in_seq[idx])
        if total >= batch_size:
            yield batch
            batch, total = [], 0
    if batch and not drop_last:
        yield batch


def collate_scenes(data, indices):
    """obs (n,T,2), pred (n,T,2), scene_mask (n,n) bool, seq_start_end (len(indices),2)
    (traj_collate_fn, utils/dataloader.py:37-65)."""
    obs = torch.cat([data[i][0] for i in indices], dim=0)
    pred = torch.cat([data[i][1] for i in indices], dim=0)
    lens = [data[i][0].shape[0] for i in indices]
    ends = np.cumsum(lens)
    sse = torch.tensor([[int(e - l), int(e)] for e, l in zip(ends, lens)], dtype=torch.long)
    mask = torch.zeros((int(ends[-1]), int(ends[-1])), dtype=torch.bool)
    for s, e in sse.tolist():
        mask[s:e, s:e] = True
    return obs, pred, mask, sse
