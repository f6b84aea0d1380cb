"""Own ETH/UCY loader vs the windows the reference's loader produced (tests/golden/data/*.npz were
written by utils/dataloader.py's TrajectoryDataset through tools/make_golden.py)."""
import os

import numpy as np
import torch

from . import _golden as G

RAW = os.path.join(G.GOLDEN, "raw")


def test_eth_test_split_identical_to_reference():
    from eigentrajectory_amd.data import TrajectoryData
    d = TrajectoryData(os.path.join(RAW, "eth_test"))
    obs, pred, sse = G.dataset("eth", "test")
    assert len(d) == 70 and d.obs_traj.shape == (181, 8, 2) and d.pred_traj.shape == (181, 12, 2)
    assert np.array_equal(d.obs_traj.numpy(), obs) and np.array_equal(d.pred_traj.numpy(), pred)
    assert np.array_equal(np.asarray(d.seq_start_end), sse)
    assert d.obs_traj.dtype == torch.float32 and d.obs_traj.is_contiguous()
    assert (d.num_peds_in_seq > 1).all()  # windows with a single pedestrian are dropped (strict >)


def test_multi_file_split_same_scenes_as_reference():
    """eth/val has 7 files; the reference concatenates them in os.listdir order, this loader by name:
    the scenes must be the same set, scene by scene."""
    from eigentrajectory_amd.data import TrajectoryData
    d = TrajectoryData(os.path.join(RAW, "eth_val"))
    obs, pred, sse = G.dataset("eth", "val")
    assert d.obs_traj.shape == obs.shape and len(d) == len(sse)

    def scene_keys(o, p, bounds):
        # "+ 0.0" folds -0.0 (np.around of a tiny negative) into +0.0: the fixtures store integers
        return sorted((np.concatenate([o[s:e], p[s:e]], axis=1) + np.float32(0)).tobytes() for s, e in bounds)
    assert scene_keys(d.obs_traj.numpy(), d.pred_traj.numpy(), d.seq_start_end) == scene_keys(obs, pred, sse)


def test_batching_and_collate():
    from eigentrajectory_amd.data import TrajectoryData, collate_scenes, scene_batches
    d = TrajectoryData(os.path.join(RAW, "eth_test"))
    batches = list(scene_batches(d.num_peds_in_seq, batch_size=16))
    assert sum(len(b) for b in batches) == len(d)
    for b in batches[:-1]:
        assert d.num_peds_in_seq[b].sum() >= 16 and d.num_peds_in_seq[b[:-1]].sum() < 16
    assert len(list(scene_batches(d.num_peds_in_seq, 16, drop_last=True))) in (len(batches), len(batches) - 1)
    g = torch.Generator().manual_seed(0)
    shuffled = [i for b in scene_batches(d.num_peds_in_seq, 16, shuffle=True, generator=g) for i in b]
    assert sorted(shuffled) == list(range(len(d))) and shuffled != list(range(len(d)))
    obs, pred, mask, sse = collate_scenes(d, batches[0])
    n = int(d.num_peds_in_seq[batches[0]].sum())
    assert obs.shape == (n, 8, 2) and pred.shape == (n, 12, 2) and mask.shape == (n, n) and sse[-1, 1] == n
    assert mask[0, 0] and mask.sum() == sum(int(k) ** 2 for k in d.num_peds_in_seq[batches[0]])
