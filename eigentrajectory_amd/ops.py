"""Functional front-end of the HIP kernels (thin: argument marshalling + autograd glue).

Each function enqueues one or two kernels of libetamd.so on the current HIP
stream and returns device tensors; none of them synchronises except
``kmeans_fit`` (the reference synchronises every iteration there,
EigenTrajectory/kmeans.py:239).  Inputs on the CPU are moved to the current HIP
device first; without a HIP device every function raises (no CPU fallback).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from ._lib import MODE_IDENTITY, MODE_MOVING, MODE_SPLIT, MODE_STATIC  # noqa: F401  (re-exported)


def _dev_args(device, *tensors):
    return [L.on_device(t, device) for t in tensors]


# ------------------------------------------------------------------------------- TrajNorm
def norm_params(obs, want_ori=True, want_rot=True, want_sca=True):
    """normalizer.py:17-29 -> traj_ori (N,1,2), traj_rot (N,2,2), traj_sca (N,1,1) (None where not wanted)."""
    dev = L.require_device(obs)
    (obs,) = _dev_args(dev, obs)
    n, t, _ = obs.shape
    ori = torch.empty((n, 1, 2), device=dev) if want_ori else None
    rot = torch.empty((n, 2, 2), device=dev) if want_rot else None
    sca = torch.empty((n, 1, 1), device=dev) if want_sca else None
    L.check(L.lib().et_norm_params(L.ptr(obs), L.i64(n), t, L.ptr(ori), L.ptr(rot), L.ptr(sca), L.stream(dev)),
            "et_norm_params")
    return ori, rot, sca


def norm_params_from_nrm(nrm, want_ori=True, want_rot=True, want_sca=True):
    """Same tensors from the compact state nrm (4,N) cached by :func:`norm_project`."""
    dev = L.require_device(nrm)
    (nrm,) = _dev_args(dev, nrm)
    n = nrm.shape[1]
    ori = torch.empty((n, 1, 2), device=dev) if want_ori else None
    rot = torch.empty((n, 2, 2), device=dev) if want_rot else None
    sca = torch.empty((n, 1, 1), device=dev) if want_sca else None
    L.check(L.lib().et_norm_params_from_nrm(L.ptr(nrm), L.i64(n), L.ptr(ori), L.ptr(rot), L.ptr(sca), L.stream(dev)),
            "et_norm_params_from_nrm")
    return ori, rot, sca


def _traj_transform(which, traj, ori, rot, sca):
    """which: "normalize" | "denormalize" -> et_normalize / et_denormalize on contiguous device tensors."""
    n, t, _ = traj.shape
    out = torch.empty_like(traj)
    fn = L.lib().et_normalize if which == "normalize" else L.lib().et_denormalize
    L.check(fn(L.ptr(traj), L.i64(n), t, L.ptr(ori), L.ptr(rot), L.ptr(sca), L.ptr(out), L.stream(traj.device)),
            "et_" + which)
    return out


class _TrajTransform(torch.autograd.Function):
    """normalizer.py:42-62 are ordinary differentiable torch ops in the reference; here the transform is a kernel,
    so its backward is spelled out.  Differentiable w.r.t. the trajectory only (the parameters come from
    ``calculate_params`` on the observations and carry no gradient path to a predictor):
        normalize    y = ((x - o) @ R) * s      dx = (g * s) @ R^T   = denormalize(g; no origin, R, 1/s)
        denormalize  y = (x / s) @ R^T + o      dx = (g @ R) / s     = normalize(g; no origin, R, 1/s)"""

    @staticmethod
    def forward(ctx, traj, ori, rot, sca, which):
        ctx.saved = (rot, sca, which)
        return _traj_transform(which, traj, ori, rot, sca)

    @staticmethod
    def backward(ctx, grad_out):
        rot, sca, which = ctx.saved
        inv = None if sca is None else (1.0 / sca).contiguous()
        other = "denormalize" if which == "normalize" else "normalize"
        return _traj_transform(other, grad_out.contiguous().float(), None, rot, inv), None, None, None, None


def _traj_op(which, traj, ori, rot, sca):
    dev = L.require_device(traj, ori, rot, sca)
    if traj.device != dev or traj.dtype != torch.float32 or not traj.is_contiguous():
        traj = traj.to(device=dev, dtype=torch.float32).contiguous()  # differentiable
    ori, rot, sca = _dev_args(dev, ori, rot, sca)
    return _TrajTransform.apply(traj, ori, rot, sca, which)


def normalize(traj, ori=None, rot=None, sca=None):
    """normalizer.py:42-51 with explicit parameter tensors (None = that step is off); autograd w.r.t. ``traj``."""
    return _traj_op("normalize", traj, ori, rot, sca)


def denormalize(traj, ori=None, rot=None, sca=None):
    """normalizer.py:53-62; autograd w.r.t. ``traj``."""
    return _traj_op("denormalize", traj, ori, rot, sca)


# ---------------------------------------------------------------------------- projection
def norm_project(obs, pred, U_obs_m, U_pred_m, U_obs_s, U_pred_s, mode, static_dist=0.0,
                 want_nrm=True, want_flag=True, want_obs=True, want_pose=False):
    """descriptor.py:144-160 fused with model.py:73-90.

    Returns (C_obs (k,N) | None, C_pred (k,N) | None, nrm (4,N) | None, flag (N,) uint8 | None) and, with ``want_pose``, a
    fifth element pose (5,N): the normaliser as :func:`anchor_reconstruct_metrics` consumes it (origin, rotation x scale,
    1 / scale with the moving / static decision in its sign).
    """
    dev = L.require_device(obs)
    obs, pred, U_obs_m, U_pred_m, U_obs_s, U_pred_s = _dev_args(dev, obs, pred, U_obs_m, U_pred_m, U_obs_s, U_pred_s)
    n, t_obs, _ = obs.shape
    us_obs = U_obs_m if U_obs_m is not None else U_obs_s
    us_pred = U_pred_m if U_pred_m is not None else U_pred_s
    k = (us_obs if us_obs is not None else us_pred).shape[1]
    t_pred = pred.shape[1] if pred is not None else (us_pred.shape[0] // 2 if us_pred is not None else 1)
    c_obs = torch.empty((k, n), device=dev) if (want_obs and us_obs is not None) else None
    c_pred = torch.empty((k, n), device=dev) if pred is not None else None
    nrm = torch.empty((4, n), device=dev) if want_nrm else None
    flag = torch.empty((n,), device=dev, dtype=torch.uint8) if want_flag else None
    if want_pose:
        pose = torch.empty((5, n), device=dev)
        L.check(L.lib().et_norm_project_pose(L.ptr(obs), L.ptr(pred), L.i64(n), t_obs, t_pred, k, L.ptr(U_obs_m),
                                             L.ptr(U_pred_m), L.ptr(U_obs_s), L.ptr(U_pred_s), int(mode), L.f32(static_dist),
                                             L.ptr(c_obs), L.ptr(c_pred), L.ptr(nrm), L.ptr(flag), L.ptr(pose), L.stream(dev)),
                "et_norm_project_pose")
        return c_obs, c_pred, nrm, flag, pose
    L.check(L.lib().et_norm_project(L.ptr(obs), L.ptr(pred), L.i64(n), t_obs, t_pred, k, L.ptr(U_obs_m),
                                    L.ptr(U_pred_m), L.ptr(U_obs_s), L.ptr(U_pred_s), int(mode), L.f32(static_dist),
                                    L.ptr(c_obs), L.ptr(c_pred), L.ptr(nrm), L.ptr(flag), L.stream(dev)),
            "et_norm_project")
    return c_obs, c_pred, nrm, flag


# ------------------------------------------------------------------------- reconstruction
def _reconstruct_fwd(Cc, obs, nrm, A_m, A_s, U_m, U_s, mode, static_dist, t_obs):
    dev = Cc.device
    k, n, s = Cc.shape
    u = U_m if U_m is not None else U_s
    t_pred = u.shape[0] // 2
    out = torch.empty((s, n, t_pred, 2), device=dev)
    L.check(L.lib().et_anchor_reconstruct_fwd(L.ptr(Cc), L.i64(n), s, k, t_obs, t_pred, L.ptr(obs), L.ptr(nrm),
                                              L.ptr(A_m), L.ptr(A_s), L.ptr(U_m), L.ptr(U_s), int(mode),
                                              L.f32(static_dist), L.ptr(out), L.stream(dev)),
            "et_anchor_reconstruct_fwd")
    return out


def _reconstruct_bwd(dtraj, obs, nrm, U_m, U_s, mode, static_dist, t_obs):
    dev = dtraj.device
    s, n, t_pred, _ = dtraj.shape
    u = U_m if U_m is not None else U_s
    k = u.shape[1]
    dC = torch.empty((k, n, s), device=dev)
    L.check(L.lib().et_anchor_reconstruct_bwd(L.ptr(dtraj), L.i64(n), s, k, t_obs, t_pred, L.ptr(obs), L.ptr(nrm),
                                              L.ptr(U_m), L.ptr(U_s), int(mode), L.f32(static_dist), L.ptr(dC),
                                              L.stream(dev)), "et_anchor_reconstruct_bwd")
    return dC


class _AnchorReconstruct(torch.autograd.Function):
    """Differentiable w.r.t. C only: U, anchors and the normaliser state are detached in the
    reference too (descriptor.py:72,87; anchor.py:87)."""

    @staticmethod
    def forward(ctx, Cc, obs, nrm, A_m, A_s, U_m, U_s, mode, static_dist, t_obs):
        ctx.saved = (obs, nrm, U_m, U_s, mode, static_dist, t_obs)
        return _reconstruct_fwd(Cc, obs, nrm, A_m, A_s, U_m, U_s, mode, static_dist, t_obs)

    @staticmethod
    def backward(ctx, grad_out):
        obs, nrm, U_m, U_s, mode, static_dist, t_obs = ctx.saved
        dC = _reconstruct_bwd(grad_out.contiguous().float(), obs, nrm, U_m, U_s, mode, static_dist, t_obs)
        return (dC,) + (None,) * 9


def anchor_reconstruct(Cc, A_m, A_s, U_m, U_s, mode, static_dist=0.0, *, obs=None, nrm=None, t_obs=8):
    """anchor.py:76-88 + descriptor.py:162-176: C (k,N,S) -> (S,N,T_pred,2); autograd w.r.t. C.

    The normaliser state comes from ``nrm`` (4,N) (as returned by :func:`norm_project`) or is
    recomputed from ``obs`` (N,T_obs,2).
    """
    dev = L.require_device(Cc)
    if Cc.device != dev or Cc.dtype != torch.float32 or not Cc.is_contiguous():
        Cc = Cc.to(device=dev, dtype=torch.float32).contiguous()  # differentiable
    obs, nrm, A_m, A_s, U_m, U_s = _dev_args(dev, obs, nrm, A_m, A_s, U_m, U_s)
    if obs is not None:
        t_obs = obs.shape[1]
    return _AnchorReconstruct.apply(Cc, obs, nrm, A_m, A_s, U_m, U_s, int(mode), float(static_dist), int(t_obs))


def anchor_reconstruct_metrics(Cc, gt, A_m, A_s, U_m, U_s, mode, static_dist=0.0, *, obs=None, nrm=None, pose=None, t_obs=8):
    """Best-of-S ADE / FDE per pedestrian (utils/metrics.py:73-102) fused into the reconstruction:
    C (k,N,S), gt (N,T_pred,2) -> ade (N,), fde (N,).  No autograd (evaluation form).

    ``pose`` (5,N) from ``norm_project(..., want_pose=True)`` (same mode and static_dist): the matrix-core kernel
    (T_pred = 12, k = 6, 12 <= S <= 64) takes the normaliser from it; any other shape needs ``nrm`` or ``obs`` as well."""
    dev = L.require_device(Cc)
    Cc, gt, obs, nrm, A_m, A_s, U_m, U_s, pose = _dev_args(dev, Cc, gt, obs, nrm, A_m, A_s, U_m, U_s, pose)
    k, n, s = Cc.shape
    if obs is not None:
        t_obs = obs.shape[1]
    t_pred = gt.shape[1]
    ade = torch.empty((n,), device=dev)
    fde = torch.empty((n,), device=dev)
    L.check(L.lib().et_anchor_reconstruct_metrics_pose(L.ptr(Cc), L.i64(n), s, k, int(t_obs), t_pred, L.ptr(obs), L.ptr(nrm),
                                                       L.ptr(pose), L.ptr(A_m), L.ptr(A_s), L.ptr(U_m), L.ptr(U_s), int(mode),
                                                       L.f32(static_dist), L.ptr(gt), L.ptr(ade), L.ptr(fde), L.stream(dev)),
            "et_anchor_reconstruct_metrics_pose")
    return ade, fde


# ------------------------------------------------------------------------------------ fit
def fit_gram(obs, pred, mode, static_dist=0.0, which=1):
    """Gram matrices (fp64) of the normalised rows routed to descriptor ``which`` + their count (int64, device)."""
    dev = L.require_device(obs)
    obs, pred = _dev_args(dev, obs, pred)
    n, t_obs, _ = obs.shape
    t_pred = pred.shape[1]
    g_obs = torch.empty((2 * t_obs, 2 * t_obs), device=dev, dtype=torch.float64)
    g_pred = torch.empty((2 * t_pred, 2 * t_pred), device=dev, dtype=torch.float64)
    count = torch.empty((1,), device=dev, dtype=torch.int64)  # (written by the finish kernel, or zeroed for n == 0: no fill)
    ws_bytes = L.lib().et_fit_gram_workspace_bytes(L.i64(n), t_obs, t_pred)
    ws = torch.empty((max(ws_bytes, 8),), device=dev, dtype=torch.uint8)
    L.check(L.lib().et_fit_gram(L.ptr(obs), L.ptr(pred), L.i64(n), t_obs, t_pred, int(mode), L.f32(static_dist),
                                int(which), L.ptr(g_obs), L.ptr(g_pred), L.ptr(count), L.ptr(ws),
                                C.c_size_t(ws.numel()), L.stream(dev)), "et_fit_gram")
    return g_obs, g_pred, count


_FIT_WS_BYTES = {}  # (N, T_obs, T_pred) -> et_fit_descriptor_workspace_bytes


def fit_descriptor(obs, pred, k, mode, static_dist=0.0, which=1, want_gram=False):
    """descriptor.py:116-142 for one descriptor in ONE call (et_fit_descriptor: the Gram kernel, its partial reduction, and
    one launch that assembles and solves both eigenproblems).  -> (U_obs (2T_obs,k), U_pred (2T_pred,k), sigma_obs (k), sigma_pred (k),
    count (int64 device tensor)) and, with ``want_gram``, + (G_obs, G_pred) -- the same bits as :func:`fit_gram` +
    :func:`eigh_topk_batch`."""
    dev = L.require_device(obs)
    # (this call opens the bench step on an idle device: every microsecond of host time in front of its first launch is a
    # microsecond of the fit stage -- tensors that are already fp32 / contiguous / on the device pass as they are, the
    # workspace size is asked once per shape, one allocation holds the small outputs, the arguments cross as plain ints)
    if not (obs.device == dev and obs.dtype == torch.float32 and obs.is_contiguous() and not obs.requires_grad):
        (obs,) = _dev_args(dev, obs)
    if not (pred.device == dev and pred.dtype == torch.float32 and pred.is_contiguous() and not pred.requires_grad):
        (pred,) = _dev_args(dev, pred)
    n, t_obs, _ = obs.shape
    t_pred = pred.shape[1]
    k = int(k)
    do, dp = 2 * t_obs, 2 * t_pred
    key = (n, t_obs, t_pred)
    ws_bytes = _FIT_WS_BYTES.get(key)
    if ws_bytes is None:
        ws_bytes = _FIT_WS_BYTES[key] = max(int(L.lib().et_fit_descriptor_workspace_bytes(L.i64(n), t_obs, t_pred)), 8)
    n_small = (do + dp) * k + 2 * k
    n_small += n_small % 2  # (the int64 count behind the floats: 8-byte aligned)
    buf = torch.empty((n_small + 2,), device=dev)
    U_obs, U_pred = buf[:do * k].view(do, k), buf[do * k:(do + dp) * k].view(dp, k)
    s_obs, s_pred = buf[(do + dp) * k:(do + dp) * k + k], buf[(do + dp) * k + k:(do + dp) * k + 2 * k]
    count = buf[n_small:].view(torch.int64)
    g_obs = torch.empty((do, do), device=dev, dtype=torch.float64) if want_gram else None
    g_pred = torch.empty((dp, dp), device=dev, dtype=torch.float64) if want_gram else None
    ws = torch.empty((ws_bytes,), device=dev, dtype=torch.uint8)
    p_buf = buf.data_ptr()
    rc = L.fast("et_fit_descriptor")(obs.data_ptr(), pred.data_ptr(), n, t_obs, t_pred, k, int(mode), float(static_dist), int(which),
                                     p_buf, p_buf + 4 * do * k, p_buf + 4 * (do + dp) * k, p_buf + 4 * ((do + dp) * k + k),
                                     g_obs.data_ptr() if want_gram else None, g_pred.data_ptr() if want_gram else None,
                                     p_buf + 4 * n_small, ws.data_ptr(), ws_bytes, L.raw_stream(dev.index))
    if rc:
        L.check(rc, "et_fit_descriptor")
    return (U_obs, U_pred, s_obs, s_pred, count) + ((g_obs, g_pred) if want_gram else ())


def eigh_topk(G, k):
    """Top-k eigenvectors (n,k) fp32 and sigma (k,) = sqrt(eigenvalues) of a symmetric fp64 matrix."""
    dev = L.require_device(G)
    G = L.on_device(G, dev, torch.float64)
    n = G.shape[0]
    U = torch.empty((n, k), device=dev)
    sigma = torch.empty((k,), device=dev)
    L.check(L.lib().et_eigh_topk(L.ptr(G), n, int(k), L.ptr(U), L.ptr(sigma), L.stream(dev)), "et_eigh_topk")
    return U, sigma


def eigh_topk_batch(mats, k):
    """``eigh_topk`` of several symmetric fp64 matrices in one launch (one workgroup each, side by side).

    mats: sequence of (n_i, n_i) tensors; k: int or sequence of ints.  -> list of (U_i, sigma_i)."""
    mats = list(mats)
    if not mats:
        return []
    ks = [int(k)] * len(mats) if isinstance(k, int) else [int(x) for x in k]
    dev = L.require_device(mats[0])
    mats = [L.on_device(G, dev, torch.float64) for G in mats]
    outs = [(torch.empty((G.shape[0], kk), device=dev), torch.empty((kk,), device=dev)) for G, kk in zip(mats, ks)]
    b = len(mats)
    PD, PF, PI = C.c_void_p * b, C.c_void_p * b, C.c_int * b
    L.check(L.lib().et_eigh_topk_batch(b, PD(*[G.data_ptr() for G in mats]), PI(*[G.shape[0] for G in mats]), PI(*ks),
                                       PF(*[U.data_ptr() for U, _ in outs]), PF(*[s.data_ptr() for _, s in outs]),
                                       L.stream(dev)), "et_eigh_topk_batch")
    return outs


# -------------------------------------------------------------------------------- k-means
def euc_sim(a, b):
    """kmeans.py:59-76: a (d,m), b (d,n) -> (m,n), or batched a (B,d,m), b (B,d,n) -> (B,m,n) in one launch."""
    dev = L.require_device(a, b)
    a, b = _dev_args(dev, a, b)
    if a.dim() == 3:
        B, d, m = a.shape
        n = b.shape[2]
        y = torch.empty((B, m, n), device=dev)
        L.check(L.lib().et_euc_sim_batch(L.ptr(a), L.ptr(b), L.i64(B), d, L.i64(m), L.i64(n), L.ptr(y), L.stream(dev)),
                "et_euc_sim_batch")
        return y
    d, m = a.shape
    n = b.shape[1]
    y = torch.empty((m, n), device=dev)
    L.check(L.lib().et_euc_sim(L.ptr(a), L.ptr(b), d, L.i64(m), L.i64(n), L.ptr(y), L.stream(dev)), "et_euc_sim")
    return y


def kmeans_workspace(n, d, K, device):
    nbytes = L.lib().et_kmeans_workspace_bytes(L.i64(n), int(d), int(K))
    if nbytes == 0:
        raise ValueError(f"k-means dimensions out of range: d={d} (<= {L.KMEANS_MAX_D}), K={K} (<= {L.KMEANS_MAX_CLUSTERS})")
    return torch.empty((nbytes,), device=device, dtype=torch.uint8)


def kmeans_init_farthest(X, K, first_index, workspace=None):
    """kmeans.py:78-112 for one batch element: X (d,N) -> C0 (d,K)."""
    dev = L.require_device(X)
    (X,) = _dev_args(dev, X)
    d, n = X.shape
    ws = workspace if workspace is not None else kmeans_workspace(n, d, K, dev)
    c0 = torch.empty((d, K), device=dev)
    L.check(L.lib().et_kmeans_init_farthest(L.ptr(X), L.i64(n), d, int(K), L.i64(first_index), L.ptr(c0), L.ptr(ws),
                                            C.c_size_t(ws.numel()), L.stream(dev)), "et_kmeans_init_farthest")
    return c0


def kmeans_fit(X, centroids, max_iter=100, tol=1e-4, workspace=None, timing=False, trace=True):
    """kmeans.py:228-240 for one batch element from given initial centroids.

    Returns dict(centroids (d,K), labels (N,) int64, n_iter, error, inertia, trace (n_iter,2) | None, done).
    ``trace=False`` skips the per-iteration (error, inertia) record -- the reference only prints it when verbose
    (kmeans.py:236-237); the inertia of the last assignment is then evaluated once, after the loop (same bits).
    """
    dev = L.require_device(X)
    X, centroids = _dev_args(dev, X, centroids)
    d, n = X.shape
    K = centroids.shape[1]
    ws = workspace if workspace is not None else kmeans_workspace(n, d, K, dev)
    cen = centroids.clone()
    labels = torch.empty((n,), device=dev, dtype=torch.int64)
    trace_t = torch.zeros((max_iter, 2), device=dev) if trace else None
    st = L.KMeansState()
    tm = L.KMeansTiming() if timing else None
    L.check(L.lib().et_kmeans_fit(L.ptr(X), L.i64(n), d, K, int(max_iter), L.f32(tol), L.ptr(cen), L.ptr(labels),
                                  L.ptr(trace_t), C.byref(st), C.byref(tm) if timing else None, L.ptr(ws),
                                  C.c_size_t(ws.numel()), L.stream(dev)), "et_kmeans_fit")
    out = dict(centroids=cen, labels=labels, n_iter=int(st.iter), error=float(st.error), inertia=float(st.inertia),
               trace=trace_t[:int(st.iter)] if trace else None, done=bool(st.done))
    if timing:
        out["assign_ms"], out["assign_launches"] = float(tm.assign_ms), int(tm.assign_launches)
        out["first_assign_ms"], out["assign_iterations"] = float(tm.first_assign_ms), int(tm.iterations)
    return out


def kmeans_fit_batch(X, centroids, max_iter=100, tol=1e-4, want_labels=False):
    """``B`` independent Lloyd fits in one call, each stopping on its own error (the n_init initialisations of the
    sklearn recipe, anchor.py:65-71): X (d,N) shared by all problems or (B,d,N); centroids (B,d,K) initial -> final.
    Returns dict(centroids (B,d,K), labels (B,N) int64 | None, n_iter [B], error [B], inertia [B], done [B])."""
    dev = L.require_device(X)
    X, centroids = _dev_args(dev, X, centroids)
    B, d, K = centroids.shape
    n = X.shape[-1]
    x_stride = 0 if X.dim() == 2 else d * n
    nbytes = L.lib().et_kmeans_batch_workspace_bytes(L.i64(n), int(d), int(K), L.i64(B))
    if nbytes == 0:
        raise ValueError(f"k-means dimensions out of range: d={d}, K={K}")
    ws = torch.empty((nbytes,), device=dev, dtype=torch.uint8)
    cen = centroids.clone()
    labels = torch.empty((B, n), device=dev, dtype=torch.int64) if want_labels else None
    states = (L.KMeansState * B)()
    L.check(L.lib().et_kmeans_fit_batch(L.ptr(X), L.i64(x_stride), L.i64(n), int(d), int(K), L.i64(B), int(max_iter), L.f32(tol),
                                        L.ptr(cen), L.ptr(labels), states, L.ptr(ws), C.c_size_t(ws.numel()), L.stream(dev)),
            "et_kmeans_fit_batch")
    return dict(centroids=cen, labels=labels, n_iter=[int(s_.iter) for s_ in states], error=[float(s_.error) for s_ in states],
                inertia=[float(s_.inertia) for s_ in states], done=[bool(s_.done) for s_ in states])


def kmeanspp_seed_batch(X, K, uniforms):
    """``B`` greedy k-means++ seedings of the same X (d,N) side by side (the y dimension of the same 4K-1 launches):
    uniforms (B, 1 + (K-1)*n_trials) float64 -> (centers (B,d,K), indices (B,K) int64).  No host synchronisation."""
    dev = L.require_device(X)
    (X,) = _dev_args(dev, X)
    uniforms = L.on_device(uniforms, dev, torch.float64)
    d, n = X.shape
    nt = kmeanspp_trials(K)
    B = uniforms.shape[0]
    if uniforms.dim() != 2 or uniforms.shape[1] != 1 + (K - 1) * nt:
        raise ValueError(f"k-means++ seeding of {K} centres consumes {1 + (K - 1) * nt} draws per initialisation")
    nbytes = L.lib().et_kmeanspp_batch_workspace_bytes(L.i64(n), d, nt, L.i64(B))
    if nbytes == 0:
        raise ValueError(f"k-means++ dimensions out of range: N={n}, d={d}, K={K}")
    ws = torch.empty((nbytes,), device=dev, dtype=torch.uint8)
    centers = torch.empty((B, d, K), device=dev)
    indices = torch.empty((B, K), device=dev, dtype=torch.int64)
    L.check(L.lib().et_kmeanspp_seed_batch(L.ptr(X), L.i64(n), d, int(K), nt, L.ptr(uniforms), L.i64(B), L.ptr(centers),
                                           L.ptr(indices), L.ptr(ws), C.c_size_t(ws.numel()), L.stream(dev)),
            "et_kmeanspp_seed_batch")
    return centers, indices


def kmeans_joint_done(state_ptrs, n_problems, tol):
    """kmeans.py:228-240 for a batch of problems run through the step API: ``state_ptrs`` = int64 device tensor of the
    problems' state-block addresses; sets every state's ``done`` from the SUM of their errors."""
    dev = L.require_device(state_ptrs)
    L.check(L.lib().et_kmeans_joint_done(L.ptr(state_ptrs), int(n_problems), L.f32(tol), L.stream(dev)),
            "et_kmeans_joint_done")


def kmeans_predict(X, centroids, want_maxsims=True):
    """kmeans.py:143-158 / 261-272: labels (N,) int64 and max similarity (N,); batched (B,d,N), (B,d,K) -> (B,N) in one
    launch."""
    dev = L.require_device(X)
    X, centroids = _dev_args(dev, X, centroids)
    if centroids.dim() == 3:  # a batch of centroid sets: on their own points (B,d,N) or all on the same points (d,N)
        B = centroids.shape[0]
        d, n = X.shape[-2], X.shape[-1]
        labels = torch.empty((B, n), device=dev, dtype=torch.int64)
        maxsims = torch.empty((B, n), device=dev) if want_maxsims else None
        L.check(L.lib().et_kmeans_predict_batch(L.ptr(X), L.i64(d * n if X.dim() == 3 else 0), L.i64(B), L.i64(n), d,
                                                L.ptr(centroids), centroids.shape[2], L.ptr(labels), L.ptr(maxsims),
                                                L.stream(dev)), "et_kmeans_predict_batch")
        return labels, maxsims
    d, n = X.shape
    labels = torch.empty((n,), device=dev, dtype=torch.int64)
    maxsims = torch.empty((n,), device=dev) if want_maxsims else None
    L.check(L.lib().et_kmeans_predict(L.ptr(X), L.i64(n), d, L.ptr(centroids), centroids.shape[1], L.ptr(labels),
                                      L.ptr(maxsims), L.stream(dev)), "et_kmeans_predict")
    return labels, maxsims


# ------------------------------------- BatchKMeans in the reference's own summation orders (opt-in)
def _reforder_ws(n, d, K, dev):
    nbytes = L.lib().et_kmeans_reforder_workspace_bytes(L.i64(n), int(d), int(K))
    if nbytes == 0:
        raise ValueError(f"k-means dimensions out of range: d={d} (<= {L.KMEANS_MAX_D}), K={K} (<= {L.KMEANS_MAX_CLUSTERS})")
    return torch.empty((nbytes,), device=dev, dtype=torch.uint8)


def euc_sim_reference_order(a, b):
    """kmeans.py:59-76 with the norms summed in torch's own order: a (d,m), b (d,n) -> (m,n), every bit the reference's."""
    dev = L.require_device(a, b)
    a, b = _dev_args(dev, a, b)
    d, m = a.shape
    n = b.shape[1]
    y = torch.empty((m, n), device=dev)
    L.check(L.lib().et_euc_sim_reforder(L.ptr(a), L.ptr(b), d, L.i64(m), L.i64(n), L.ptr(y), L.stream(dev)),
            "et_euc_sim_reforder")
    return y


def kmeans_init_farthest_reference_order(X, K, first_index):
    """kmeans.py:78-112 literally (euc_sim against all current centroids at every step, torch's norm orders)."""
    dev = L.require_device(X)
    (X,) = _dev_args(dev, X)
    d, n = X.shape
    ws = _reforder_ws(n, d, K, dev)
    c0 = torch.empty((d, K), device=dev)
    L.check(L.lib().et_kmeans_init_farthest_reforder(L.ptr(X), L.i64(n), d, int(K), L.i64(first_index), L.ptr(c0), L.ptr(ws),
                                                     C.c_size_t(ws.numel()), L.stream(dev)), "et_kmeans_init_farthest_reforder")
    return c0


def kmeans_predict_reference_order(X, centroids):
    """kmeans.py:143-158 with torch's norm orders: labels (N,) int64, maxsims (N,)."""
    dev = L.require_device(X)
    X, centroids = _dev_args(dev, X, centroids)
    d, n = X.shape
    K = centroids.shape[1]
    ws = _reforder_ws(n, d, K, dev)
    labels = torch.empty((n,), device=dev, dtype=torch.int64)
    maxsims = torch.empty((n,), device=dev)
    L.check(L.lib().et_kmeans_predict_reforder(L.ptr(X), L.i64(n), d, L.ptr(centroids), K, L.ptr(labels), L.ptr(maxsims),
                                               L.ptr(ws), C.c_size_t(ws.numel()), L.stream(dev)), "et_kmeans_predict_reforder")
    return labels, maxsims


def kmeans_fit_reference_order(X, centroids, max_iter=100, tol=1e-4, trace=True, timing=False):
    """kmeans.py:228-240 with the cluster sums, norms and error in the reference's fp32 orders (single GPU).
    Same return dict as :func:`kmeans_fit`.  d = 6, K <= 32, N >= 1024: one launch per iteration (the parallel form of
    ATen's cascade, csrc/et_kmeans_reforder.hip, namespace fast); any other shape: the plain kernels."""
    return kmeans_fit_reference_order_batch(X[None], centroids[None], max_iter, tol, trace=trace, timing=timing)[0]


def kmeans_fit_reference_order_batch(X, centroids, max_iter=100, tol=1e-4, trace=True, timing=False):
    """BatchKMeans.fit's loop (kmeans.py:228-240) for X (l, d, N), centroids (l, d, K) in the reference's summation
    orders: ALL problems iterate in one loop and stop together on the error summed over the whole (l, d, K) tensor.
    Returns one dict per problem (all with the same n_iter / error)."""
    dev = L.require_device(X)
    X, centroids = _dev_args(dev, X, centroids)
    nb, d, n = X.shape
    K = centroids.shape[2]
    nbytes = L.lib().et_kmeans_reforder_batch_workspace_bytes(L.i64(n), int(d), int(K), L.i64(nb))
    if nbytes == 0:
        if nb > 1:
            raise NotImplementedError(f"sums='reference-order' with l = {nb} > 1 problems takes d = 6, K <= 32, 1024 <= N < 2^29, "
                                      f"l <= 64 (got d={d}, K={K}, N={n})")
        raise ValueError(f"k-means dimensions out of range: d={d} (<= {L.KMEANS_MAX_D}), K={K} (<= {L.KMEANS_MAX_CLUSTERS})")
    ws = torch.empty((nbytes,), device=dev, dtype=torch.uint8)
    cen = centroids.clone()
    labels = torch.empty((nb, n), device=dev, dtype=torch.int64)
    trace_t = torch.zeros((nb, max_iter, 2), device=dev) if trace else None
    st = (L.KMeansState * nb)()
    tm = L.KMeansTiming() if timing else None
    L.check(L.lib().et_kmeans_fit_reforder_batch(L.ptr(X), L.i64(d * n), L.i64(n), d, K, L.i64(nb), int(max_iter), L.f32(tol),
                                                 L.ptr(cen), L.ptr(labels), L.ptr(trace_t), st, C.byref(tm) if timing else None,
                                                 L.ptr(ws), C.c_size_t(ws.numel()), L.stream(dev)),
            "et_kmeans_fit_reforder_batch")
    out = []
    for b in range(nb):
        r = dict(centroids=cen[b], labels=labels[b], n_iter=int(st[b].iter), error=float(st[b].error),
                 inertia=float(st[b].inertia), trace=trace_t[b, :int(st[b].iter)] if trace else None, done=bool(st[b].done))
        if timing:
            r["timing"] = dict(loop_ms=float(tm.assign_ms), launches=int(tm.assign_launches), iterations=int(tm.iterations))
        out.append(r)
    return out


def reference_order_shard_sizes(n_total, world, d=6, K=20):
    """How to split n_total points over `world` ranks so that sums="reference-order" gives the single-GPU bits: every
    rank before the last non-empty one holds whole level-2 blocks of ATen's cascade (include/eigentraj.h,
    et_kmeans_fit_reforder_sharded); as even as that allows.  -> list of `world` sizes (trailing ranks may be empty)."""
    block = int(L.lib().et_kmeans_reforder_shard_block(L.i64(n_total), int(d), int(K)))
    if block == 0:
        raise NotImplementedError(f"sharded sums='reference-order' takes d = 6, K <= 32, 1024 <= N < 2^29 (got d={d}, K={K}, N={n_total})")
    blocks = -(-int(n_total) // block)
    per = -(-blocks // int(world))
    sizes, left = [], int(n_total)
    for _ in range(int(world)):
        take = min(left, per * block)
        sizes.append(take)
        left -= take
    return sizes


def kmeans_fit_reference_order_sharded(X_local, centroids, n_locals, rank, comm=None, max_iter=100, tol=1e-4, trace=True):
    """kmeans.py:228-240 in the reference's fp32 summation orders over points split across ranks (n_locals: every rank's
    size, see :func:`reference_order_shard_sizes`; comm: a dist.Communicator or None for one rank).  The same centroids,
    labels, error and iteration count as :func:`kmeans_fit_reference_order` on the whole array, on every rank."""
    dev = L.require_device(centroids)
    X_local, centroids = _dev_args(dev, X_local, centroids)
    d, n = X_local.shape
    K = centroids.shape[1]
    sizes = (C.c_int64 * len(n_locals))(*[int(v) for v in n_locals])
    if int(n_locals[rank]) != n:
        raise ValueError(f"n_locals[{rank}] = {n_locals[rank]} but this rank holds {n} points")
    nbytes = L.lib().et_kmeans_reforder_sharded_workspace_bytes(sizes, len(n_locals), int(rank), int(d), int(K))
    if nbytes == 0:
        raise NotImplementedError("sharded sums='reference-order': d = 6, K <= 32, 1024 <= N_total < 2^29 and whole level-2 "
                                  f"blocks on every rank before the last (got d={d}, K={K}, sizes={list(n_locals)})")
    ws = torch.empty((nbytes,), device=dev, dtype=torch.uint8)
    cen = centroids.clone()
    labels = torch.empty((n,), device=dev, dtype=torch.int64)
    trace_t = torch.zeros((max_iter, 2), device=dev) if trace else None
    st = L.KMeansState()
    L.check(L.lib().et_kmeans_fit_reforder_sharded(L.ptr(X_local) if n else None, sizes, len(n_locals), int(rank), int(d), int(K),
                                                   int(max_iter), L.f32(tol), L.ptr(cen), L.ptr(labels) if n else None,
                                                   L.ptr(trace_t), C.byref(st), L.ptr(ws), C.c_size_t(ws.numel()),
                                                   comm.handle if comm is not None else None, L.stream(dev)),
            "et_kmeans_fit_reforder_sharded")
    return dict(centroids=cen, labels=labels, n_iter=int(st.iter), error=float(st.error), inertia=float(st.inertia),
                trace=trace_t[:int(st.iter)] if trace else None, done=bool(st.done))


# ---------------------------------------------------- sklearn-recipe anchors (anchor.py:65-71)
def center_columns(X, rel_tol=1e-4):
    """KMeans.fit's pre-processing on a COPY of X (d,N): -> (X - mean (d,N), mean (d,), tol (1,) = rel_tol * mean(var))
    all on the device, in numpy's float32 reduction order (csrc/et_kmeanspp.hip)."""
    dev = L.require_device(X)
    Xc = L.on_device(X, dev).clone()
    d, n = Xc.shape
    mean = torch.empty((d,), device=dev)
    tol = torch.empty((1,), device=dev)
    ws = torch.empty((2 * L.KMEANS_MAX_D,), device=dev)
    L.check(L.lib().et_center_columns(L.ptr(Xc), L.i64(n), d, L.f32(rel_tol), L.ptr(mean), L.ptr(tol), L.ptr(ws),
                                      C.c_size_t(ws.numel() * 4), L.stream(dev)), "et_center_columns")
    return Xc, mean, tol


def kmeanspp_trials(K):
    """sklearn's number of candidates per centre: 2 + floor(ln K)."""
    import math
    return 2 + int(math.log(K))


def kmeanspp_seed(X, K, uniforms, workspace=None):
    """Greedy k-means++ seeding of X (d,N) with the float64 draws ``uniforms`` (1 + (K-1)*n_trials,) on the device
    -> (centers (d,K), indices (K,) int64).  No host synchronisation."""
    dev = L.require_device(X)
    (X,) = _dev_args(dev, X)
    uniforms = L.on_device(uniforms, dev, torch.float64)
    d, n = X.shape
    nt = kmeanspp_trials(K)
    if uniforms.numel() != 1 + (K - 1) * nt:
        raise ValueError(f"k-means++ seeding of {K} centres consumes {1 + (K - 1) * nt} draws, got {uniforms.numel()}")
    if workspace is None:
        nbytes = L.lib().et_kmeanspp_workspace_bytes(L.i64(n), d, nt)
        if nbytes == 0:
            raise ValueError(f"k-means++ dimensions out of range: N={n}, d={d}, K={K}")
        workspace = torch.empty((nbytes,), device=dev, dtype=torch.uint8)
    centers = torch.empty((d, K), device=dev)
    indices = torch.empty((K,), device=dev, dtype=torch.int64)
    L.check(L.lib().et_kmeanspp_seed(L.ptr(X), L.i64(n), d, int(K), nt, L.ptr(uniforms), L.ptr(centers), L.ptr(indices),
                                     L.ptr(workspace), C.c_size_t(workspace.numel()), L.stream(dev)), "et_kmeanspp_seed")
    return centers, indices


class KMeansShard:
    """Step-wise Lloyd iteration on one shard of the points (the sharded / multi-GPU form).

    ``scan`` -> [all-reduce MAX of ``state_f64[0]`` and ``state_i64[7]``] -> ``begin`` ->
    repeat { ``assign`` -> [all-reduce SUM of the int64 partials] -> ``update`` } -> ``labels``.
    Everything stays on the device; ``state`` is read back only when the caller asks.
    """

    def __init__(self, X, K):
        self.dev = L.require_device(X)
        self.X = L.on_device(X, self.dev)
        self.d, self.n = self.X.shape
        self.K = int(K)
        self.ws = kmeans_workspace(self.n, self.d, self.K, self.dev)
        self.state = torch.zeros((L.STATE_BYTES // 8,), device=self.dev, dtype=torch.int64)
        self.partials = torch.zeros((L.lib().et_kmeans_partials_len(self.d, self.K),), device=self.dev,
                                    dtype=torch.int64)
        self.labels_u8 = torch.zeros((max(self.n, 1) + 3,), device=self.dev, dtype=torch.uint8)
        self.best = torch.empty((max(self.n, 1),), device=self.dev)
        self.cand = torch.zeros((8 + 4 * L.KMEANS_MAX_D,), device=self.dev, dtype=torch.uint8)

    @property
    def state_f64(self):
        return self.state.view(torch.float64)

    def scan(self):
        L.check(L.lib().et_kmeans_scan(L.ptr(self.X), L.i64(self.n), self.d, L.ptr(self.state), L.stream(self.dev)),
                "et_kmeans_scan")

    def begin(self, n_total, centroids):
        L.check(L.lib().et_kmeans_begin(L.ptr(self.state), L.i64(n_total), L.ptr(centroids), self.d, self.K,
                                        L.stream(self.dev)), "et_kmeans_begin")

    def init_step(self, i, C0, index_base):
        """candidate record of farthest-first step i: uint8 tensor {key u64, d floats}."""
        L.check(L.lib().et_kmeans_init_step(L.ptr(self.X), L.i64(self.n), self.d, self.K, int(i), L.ptr(C0),
                                            L.ptr(self.best), L.i64(index_base), L.ptr(self.cand), L.ptr(self.ws),
                                            C.c_size_t(self.ws.numel()), L.stream(self.dev)), "et_kmeans_init_step")
        return self.cand

    def init_select(self, cands, n_cands, stride, col, C0):
        """Column ``col`` of C0 <- the candidate record with the smallest key among ``n_cands`` gathered records."""
        L.check(L.lib().et_kmeans_init_select(L.ptr(cands), int(n_cands), int(stride), self.d, self.K, int(col), L.ptr(C0),
                                              L.stream(self.dev)), "et_kmeans_init_select")

    def post_state(self):
        """Start an asynchronous copy of the state block to pinned host memory; -> handle for ``wait_state``."""
        if not hasattr(self, "_pins"):
            self._pins, self._pin_i = [torch.empty((L.STATE_BYTES // 8,), dtype=torch.int64).pin_memory() for _ in range(4)], 0
        host = self._pins[self._pin_i % 4]
        self._pin_i += 1
        host.copy_(self.state, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        return host, ev

    def wait_state(self, handle):
        host, ev = handle
        ev.synchronize()
        return L.KMeansState.from_buffer_copy(host.numpy().tobytes())

    def gather_point(self, local_index):
        pt = torch.empty((self.d,), device=self.dev)
        L.check(L.lib().et_kmeans_gather_point(L.ptr(self.X), L.i64(self.n), self.d, L.i64(local_index), L.ptr(pt),
                                               L.stream(self.dev)), "et_kmeans_gather_point")
        return pt

    def assign(self, centroids, given_labels=None):
        L.check(L.lib().et_kmeans_assign_accumulate(L.ptr(self.X), L.i64(self.n), self.d, self.K, L.ptr(self.state),
                                                    L.ptr(centroids), L.ptr(given_labels), L.ptr(self.labels_u8),
                                                    L.ptr(self.partials), L.ptr(self.ws), C.c_size_t(self.ws.numel()),
                                                    L.stream(self.dev)), "et_kmeans_assign_accumulate")
        return self.partials

    def update(self, partials, centroids, tol, trace=None):
        L.check(L.lib().et_kmeans_update(L.ptr(self.state), L.ptr(partials), self.d, self.K, L.f32(tol),
                                         L.ptr(centroids), L.ptr(trace), L.stream(self.dev)), "et_kmeans_update")

    def labels(self):
        out = torch.empty((self.n,), device=self.dev, dtype=torch.int64)
        L.check(L.lib().et_kmeans_labels_i64(L.ptr(self.labels_u8), L.i64(self.n), L.ptr(out), L.stream(self.dev)),
                "et_kmeans_labels_i64")
        return out

    def read_state(self):
        """Blocking read-back of the state block -> _lib.KMeansState."""
        host = self.state.cpu().numpy().tobytes()
        return L.KMeansState.from_buffer_copy(host)
