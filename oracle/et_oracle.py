"""ctypes/numpy front-end of the CPU oracle (oracle/et_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of et_oracle.c.  Imported by
tests/, by ``__graft_entry__.smoke()`` and by ``bench.py``'s cpu_baseline leg;
never by ``eigentrajectory_amd``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libet_oracle.so")
_lib = None

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "et_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libet_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _check(rc, what):
    if rc != 0:
        raise ValueError(f"oracle {what} failed with status {rc}")


# ------------------------------------------------------------------ TrajNorm
def norm_params(obs, use_sca=True):
    obs = _f32(obs)
    n, t, _ = obs.shape
    ori = np.empty((n, 1, 2), np.float32)
    rot = np.empty((n, 2, 2), np.float32)
    sca = np.empty((n, 1, 1), np.float32) if use_sca else None
    _check(lib().eto_norm_params(_p(obs, _f32p), C.c_int64(n), t, int(use_sca), _p(ori, _f32p), _p(rot, _f32p),
                                 _p(sca, _f32p)), "norm_params")
    return ori, rot, sca


def normalize(obs, traj, use_sca=True):
    obs, traj = _f32(obs), _f32(traj)
    out = np.empty_like(traj)
    _check(lib().eto_normalize(_p(obs, _f32p), _p(traj, _f32p), C.c_int64(obs.shape[0]), obs.shape[1], traj.shape[1],
                               int(use_sca), _p(out, _f32p)), "normalize")
    return out


def denormalize(obs, traj_norm, use_sca=True):
    obs, traj_norm = _f32(obs), _f32(traj_norm)
    out = np.empty_like(traj_norm)
    _check(lib().eto_denormalize(_p(obs, _f32p), _p(traj_norm, _f32p), C.c_int64(obs.shape[0]), obs.shape[1],
                                 traj_norm.shape[1], int(use_sca), _p(out, _f32p)), "denormalize")
    return out


def moving_flags(obs, static_dist):
    obs = _f32(obs)
    flag = np.empty((obs.shape[0],), np.uint8)
    _check(lib().eto_moving_flags(_p(obs, _f32p), C.c_int64(obs.shape[0]), obs.shape[1], C.c_float(static_dist),
                                  _p(flag, _u8p)), "moving_flags")
    return flag.astype(bool)


# -------------------------------------------------------------- ETDescriptor
def norm_project(obs, pred, U_obs_m, U_pred_m, U_obs_s, U_pred_s, mode, static_dist=0.0):
    """-> C_obs (k,N), C_pred (k,N)|None, nrm (4,N), flag (N,) uint8."""
    obs = _f32(obs)
    n, t_obs, _ = obs.shape
    pred = None if pred is None else _f32(pred)
    t_pred = pred.shape[1] if pred is not None else (U_pred_m if U_pred_m is not None else U_pred_s).shape[0] // 2
    us = [None if u is None else _f32(u) for u in (U_obs_m, U_pred_m, U_obs_s, U_pred_s)]
    k = next(u for u in us if u is not None).shape[1]
    c_obs = np.empty((k, n), np.float32) if (us[0] is not None or us[2] is not None) else None
    c_pred = np.empty((k, n), np.float32) if pred is not None else None
    nrm = np.empty((4, n), np.float32)
    flag = np.empty((n,), np.uint8)
    _check(lib().eto_norm_project(_p(obs, _f32p), _p(pred, _f32p), C.c_int64(n), t_obs, t_pred, k,
                                  _p(us[0], _f32p), _p(us[1], _f32p), _p(us[2], _f32p), _p(us[3], _f32p),
                                  int(mode), C.c_float(static_dist), _p(c_obs, _f32p), _p(c_pred, _f32p),
                                  _p(nrm, _f32p), _p(flag, _u8p)), "norm_project")
    return c_obs, c_pred, nrm, flag


def anchor_reconstruct(Cc, obs, A_m, A_s, U_m, U_s, mode, static_dist=0.0):
    """C (k,N,S) -> (S,N,T_pred,2)."""
    Cc, obs = _f32(Cc), _f32(obs)
    k, n, s = Cc.shape
    us = [None if u is None else _f32(u) for u in (A_m, A_s, U_m, U_s)]
    t_pred = next(u for u in us[2:] if u is not None).shape[0] // 2
    out = np.empty((s, n, t_pred, 2), np.float32)
    _check(lib().eto_anchor_reconstruct(_p(Cc, _f32p), C.c_int64(n), s, k, obs.shape[1], t_pred, _p(obs, _f32p),
                                        _p(us[0], _f32p), _p(us[1], _f32p), _p(us[2], _f32p), _p(us[3], _f32p),
                                        int(mode), C.c_float(static_dist), _p(out, _f32p)), "anchor_reconstruct")
    return out


def anchor_reconstruct_bwd(dtraj, obs, U_m, U_s, mode, static_dist=0.0):
    """dtraj (S,N,T,2) -> dC (k,N,S)."""
    dtraj, obs = _f32(dtraj), _f32(obs)
    s, n, t_pred, _ = dtraj.shape
    us = [None if u is None else _f32(u) for u in (U_m, U_s)]
    k = next(u for u in us if u is not None).shape[1]
    out = np.empty((k, n, s), np.float32)
    _check(lib().eto_anchor_reconstruct_bwd(_p(dtraj, _f32p), C.c_int64(n), s, k, obs.shape[1], t_pred,
                                            _p(obs, _f32p), _p(us[0], _f32p), _p(us[1], _f32p), int(mode),
                                            C.c_float(static_dist), _p(out, _f32p)), "anchor_reconstruct_bwd")
    return out


def fit_gram(obs, pred, mode, static_dist=0.0, which=1):
    obs, pred = _f32(obs), _f32(pred)
    do, dp = 2 * obs.shape[1], 2 * pred.shape[1]
    g_obs = np.empty((do, do), np.float64)
    g_pred = np.empty((dp, dp), np.float64)
    cnt = C.c_int64(0)
    _check(lib().eto_fit_gram(_p(obs, _f32p), _p(pred, _f32p), C.c_int64(obs.shape[0]), obs.shape[1], pred.shape[1],
                              int(mode), C.c_float(static_dist), int(which), _p(g_obs, _f64p), _p(g_pred, _f64p),
                              C.byref(cnt)), "fit_gram")
    return g_obs, g_pred, cnt.value


def eigh_topk(G, k):
    G = np.ascontiguousarray(G, dtype=np.float64)
    n = G.shape[0]
    U = np.empty((n, k), np.float32)
    sigma = np.empty((k,), np.float32)
    _check(lib().eto_eigh_topk(_p(G, _f64p), n, k, _p(U, _f32p), _p(sigma, _f32p)), "eigh_topk")
    return U, sigma


# --------------------------------------------------------------- BatchKMeans
def _ref(name, reference_order):
    return getattr(lib(), name + ("_ref" if reference_order else ""))


def inner_sum(v):
    """torch's fp32 order for a contiguous full reduction (kmeans.py:50 ``diff.sum()``)."""
    v = _f32(v).ravel()
    f = lib().eto_inner_sum_f32
    f.restype = C.c_float
    return float(f(_p(v, _f32p), C.c_int64(v.size)))


def euc_sim(a, b, reference_order=False):
    a, b = _f32(a), _f32(b)
    d, m = a.shape
    n = b.shape[1]
    y = np.empty((m, n), np.float32)
    _check(_ref("eto_euc_sim", reference_order)(_p(a, _f32p), _p(b, _f32p), d, C.c_int64(m), C.c_int64(n), _p(y, _f32p)),
           "euc_sim")
    return y


def kmeans_assign(X, Cn, reference_order=False):
    X, Cn = _f32(X), _f32(Cn)
    d, n = X.shape
    labels = np.empty((n,), np.int64)
    maxsims = np.empty((n,), np.float32)
    _check(_ref("eto_kmeans_assign", reference_order)(_p(X, _f32p), C.c_int64(n), d, _p(Cn, _f32p), Cn.shape[1],
                                                      _p(labels, _i64p), _p(maxsims, _f32p)), "kmeans_assign")
    return labels, maxsims


def kmeans_init_farthest(X, K, first_index, reference_order=False):
    X = _f32(X)
    d, n = X.shape
    c0 = np.empty((d, K), np.float32)
    idx = np.empty((K,), np.int64)
    _check(_ref("eto_kmeans_init_farthest", reference_order)(_p(X, _f32p), C.c_int64(n), d, K, C.c_int64(first_index),
                                                             _p(c0, _f32p), _p(idx, _i64p)), "kmeans_init_farthest")
    return c0, idx


def kmeans_frac_bits(max_abs, n_total):
    f = lib().eto_kmeans_frac_bits
    f.restype = C.c_int
    return f(C.c_double(max_abs), C.c_int64(n_total))


def kmeans_sim_frac_bits(max_abs_x, max_abs_c, d, n_total):
    f = lib().eto_kmeans_sim_frac_bits
    f.restype = C.c_int
    return f(C.c_double(max_abs_x), C.c_double(max_abs_c), int(d), C.c_int64(n_total))


def kmeans_assign_accumulate(X, Cn, frac, sim_frac):
    """One shard's exact partials: labels, sums (d,K) i64, counts (K) i64, sim_sum, nan_count."""
    X, Cn = _f32(X), _f32(Cn)
    d, n = X.shape
    K = Cn.shape[1]
    labels = np.empty((n,), np.int64)
    sums = np.empty((d, K), np.int64)
    counts = np.empty((K,), np.int64)
    ss, nn = C.c_int64(0), C.c_int64(0)
    _check(lib().eto_kmeans_assign_accumulate(_p(X, _f32p), C.c_int64(n), d, _p(Cn, _f32p), K, int(frac),
                                              int(sim_frac), _p(labels, _i64p), _p(sums, _i64p), _p(counts, _i64p),
                                              C.byref(ss), C.byref(nn)), "kmeans_assign_accumulate")
    return labels, sums, counts, ss.value, nn.value


def kmeans_update(sums, counts, sim_sum, nan_count, n_total, frac, sim_frac, tol, C_old):
    sums = np.ascontiguousarray(sums, np.int64)
    counts = np.ascontiguousarray(counts, np.int64)
    C_old = _f32(C_old)
    d, K = C_old.shape
    C_new = np.empty_like(C_old)
    err, ine, done = C.c_float(0), C.c_float(0), C.c_int(0)
    _check(lib().eto_kmeans_update(_p(sums, _i64p), _p(counts, _i64p), C.c_int64(sim_sum), C.c_int64(nan_count),
                                   C.c_int64(n_total), d, K, int(frac), int(sim_frac), C.c_float(tol),
                                   _p(C_old, _f32p), _p(C_new, _f32p), C.byref(err), C.byref(ine), C.byref(done)),
           "kmeans_update")
    return C_new, err.value, ine.value, bool(done.value)


def kmeans_reforder_sums(X, labels, K):
    """Per-cluster member sums (d,K) fp32 in the reference's (ATen cascade) summation order, kmeans.py:180-182."""
    X = _f32(X)
    labels = np.ascontiguousarray(labels, np.int64)
    d, n = X.shape
    sums = np.empty((d, K), np.float32)
    _check(lib().eto_kmeans_reforder_sums(_p(X, _f32p), C.c_int64(n), d, int(K), _p(labels, _i64p), _p(sums, _f32p)),
           "kmeans_reforder_sums")
    return sums


def kmeans_fit(X, C_init, max_iter=100, tol=1e-4, sums="exact"):
    """-> dict(centroids (d,K), labels (N,) int64, n_iter, error, inertia, trace (n_iter,2)).
    ``sums="reference-order"``: cluster sums in the reference's fp32 summation order instead of exactly."""
    if sums not in ("exact", "reference-order"):
        raise ValueError(sums)
    X, C_init = _f32(X), _f32(C_init)
    d, n = X.shape
    K = C_init.shape[1]
    cen = np.empty((d, K), np.float32)
    labels = np.empty((n,), np.int64)
    trace = np.zeros((max_iter, 2), np.float32)
    it, err, ine = C.c_int(0), C.c_float(0), C.c_float(0)
    fn = lib().eto_kmeans_fit if sums == "exact" else lib().eto_kmeans_fit_reforder
    _check(fn(_p(X, _f32p), C.c_int64(n), d, K, _p(C_init, _f32p), int(max_iter), C.c_float(tol), _p(cen, _f32p),
              _p(labels, _i64p), C.byref(it), C.byref(err), C.byref(ine), _p(trace, _f32p)), "kmeans_fit")
    return dict(centroids=cen, labels=labels, n_iter=it.value, error=err.value, inertia=ine.value,
                trace=trace[:it.value].copy())


def kmeans_fit_batch_reference_order(Xs, C_inits, max_iter=100, tol=1e-4):
    """BatchKMeans.fit on l problems in the reference's summation orders (kmeans.py:228-240): one loop, one error over the
    whole (l, d, K) tensor in ATen's inner-sum order.  -> dict(centroids (l,d,K), labels (l,N), n_iter, error,
    inertia (l,), trace (n_iter,2) = (joint error, mean inertia))."""
    Xs, C_inits = _f32(np.stack(Xs)), _f32(np.stack(C_inits))
    l, d, n = Xs.shape
    K = C_inits.shape[2]
    cen = np.empty((l, d, K), np.float32)
    labels = np.empty((l, n), np.int64)
    ine = np.zeros((l,), np.float32)
    trace = np.zeros((max_iter, 2), np.float32)
    it, err = C.c_int(0), C.c_float(0)
    _check(lib().eto_kmeans_fit_reforder_batch(_p(Xs, _f32p), C.c_int64(n), d, K, l, _p(C_inits, _f32p), int(max_iter),
                                               C.c_float(tol), _p(cen, _f32p), _p(labels, _i64p), C.byref(it), C.byref(err),
                                               _p(ine, _f32p), _p(trace, _f32p)), "kmeans_fit_reforder_batch")
    return dict(centroids=cen, labels=labels, n_iter=it.value, error=err.value, inertia=ine, trace=trace[:it.value].copy())


def kmeans_fit_batch(Xs, C_inits, max_iter=100, tol=1e-4):
    """BatchKMeans.fit on a batch of l problems (kmeans.py:228-240): ONE error -- the sum over all problems -- is
    compared with ``tol`` (kmeans.py:239), so all problems stop in the same iteration.  Restated on the per-problem C
    fit: every problem alone with a tolerance nothing meets gives its error trace; the joint loop stops at the first
    iteration whose summed error (fp64 over the problems' fp32 errors, problem order, rounded to fp32) is <= tol; the
    problems' results are their states after that many iterations.  -> list of kmeans_fit dicts."""
    l = len(Xs)
    full = [kmeans_fit(Xs[b], C_inits[b], max_iter, -1.0) for b in range(l)]
    n_iter = max_iter
    for t in range(max_iter):
        s = 0.0
        for b in range(l):
            s += float(full[b]["trace"][t, 0])
        if np.float32(s) <= np.float32(tol):  # NaN: keep going
            n_iter = t + 1
            break
    return full if n_iter == max_iter else [kmeans_fit(Xs[b], C_inits[b], n_iter, -1.0) for b in range(l)]
