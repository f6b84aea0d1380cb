"""numpy restatement (test infrastructure) of what a rank of et_kmeans_fit_reforder_sharded sends and what every rank does
with the gathered records (csrc/et_kmeans_reforder.hip: "The reference-order iteration over SHARDS"): the cluster sums of
kmeans.py:180-182 in ATen's cascade order from per-block rows.  Sequential fp32 additions are written as loops over the level
step with the chunks / groups / blocks side by side in numpy -- the same additions in the same order as the kernels."""
import numpy as np

f32 = np.float32


def level_step(n_total):
    """ATen SumKernel.cpp: 2^max(4, ceil_log2(items per lane) / 4) for the 4-lane interleaved sum of n_total terms."""
    size = n_total // 4
    cl = 1 if size <= 2 else int(size - 1).bit_length()
    return 1 << max(4, cl // 4)


def _seq_sum(a, axis):
    """sequential fp32 sum along `axis` (0 + a0 + a1 + ...), everything else side by side"""
    a = np.moveaxis(a, axis, 0)
    acc = np.zeros(a.shape[1:], f32)
    for r in range(a.shape[0]):
        acc = (acc + a[r]).astype(f32)
    return acc


def shard_record(X_local, labels_local, K, L):
    """One rank's record: per complete-or-partial level-2 block a row (d, K, 4 lanes), + the end of ITS range: T1 (the chunks
    after the last full group, level 1), T0 (the lane terms after the last full chunk, level 0), the leftover points."""
    d, n = X_local.shape
    masked = np.zeros((d, K, n), f32)  # data * mask: the terms ATen adds (zeros included)
    for j in range(K):
        masked[:, j, labels_local == j] = X_local[:, labels_local == j]
    size = n // 4
    lanes = masked[:, :, :size * 4].reshape(d, K, size, 4)  # [.., item, lane]
    full_chunks = size // L
    chunks = _seq_sum(lanes[:, :, :full_chunks * L].reshape(d, K, full_chunks, L, 4), 3)          # level 0 -> (d, K, chunk, 4)
    G = full_chunks // L
    groups = _seq_sum(chunks[:, :, :G * L].reshape(d, K, G, L, 4), 3)                               # level 1 -> (d, K, group, 4)
    n_blk = -(-G // L)
    rows = np.zeros((n_blk, d, K, 4), f32)
    for b in range(n_blk):                                                                          # level 2 (a partial last block as it stands)
        rows[b] = _seq_sum(groups[:, :, b * L:min((b + 1) * L, G)], 2)
    T1 = _seq_sum(chunks[:, :, G * L:], 2)                                                          # partial group
    T0 = _seq_sum(lanes[:, :, full_chunks * L:], 2)                                                 # partial chunk
    return dict(rows=rows, full_rows=G // L, T1=T1, T0=T0, left=masked[:, :, size * 4:])


def replay(records, tail_rank):
    """What every rank does with the gathered records: level 3 over the ranks' complete blocks in rank order, the tail rank's
    partial block / group / chunk, the N mod 4 leftover terms onto lane 0, the lanes -> (d, K) sums."""
    d, K = records[tail_rank]["T1"].shape[:2]
    a3 = np.zeros((d, K, 4), f32)
    p2 = np.zeros((d, K, 4), f32)
    for r, rec in enumerate(records):
        nfull = rec["full_rows"] if r == tail_rank else len(rec["rows"])
        for b in range(len(rec["rows"])):
            if b < nfull:
                a3 = (a3 + rec["rows"][b]).astype(f32)
            else:
                p2 = rec["rows"][b]
    t = records[tail_rank]
    lane = (((t["T0"] + t["T1"]).astype(f32) + p2).astype(f32) + a3).astype(f32)  # acc[0] + acc[1] + acc[2] + acc[3]
    p = lane[..., 0]
    for m in range(t["left"].shape[2]):
        p = (p + t["left"][:, :, m]).astype(f32)
    for k in range(1, 4):
        p = (p + lane[..., k]).astype(f32)
    return p
