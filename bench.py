#!/usr/bin/env python3
"""Benchmark of the descriptor hot path on MI355X (contract: see the task prompt / DESIGN.md §Measurement).

One *step* = one pass of the whole path over one batch of synthetic trajectories that are
already resident in HBM:

    fit (Gram + Jacobi) -> project (obs+pred) -> reconstruct (S=1 round trip) -> k-means
    (farthest-first init, K=20, Lloyd max_iter=100, tol=1e-4) on the pred coefficients

N = 1e7 trajectories per GPU by default (the size BASELINE.json quotes the metric on), single
descriptor with scale normalisation (SURVEY.md §8(d) "pure kernel benchmark" form).  With
--gpus N>1 every rank holds its own N trajectories (weak scaling) and the fit / k-means
exchange Gram matrices and exact per-cluster sums over RCCL.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
# algorithmic bytes per trajectory
BYTES = dict(fit=160.0, project=208.0, reconstruct=136.0, kmeans_iter=24.0, kmeans_init_step=32.0, labels=8.0)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--trajectories", dest="n", type=float, default=1e7, help="trajectories per GPU")
    ap.add_argument("--max-iter", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra stage measurements (S=20 model form, N=1e5/1e6 end-to-end, scene latency)")
    ap.add_argument("--cpu-budget", type=float, default=45.0,
                    help="seconds of host time the PyTorch-CPU baseline may use for its N=1e6 run")
    ap.add_argument("--c-port-sample", type=int, default=300_000,
                    help="trajectories of the scalar C oracle's sample (second CPU figure, ~5 s on one core)")
    return ap.parse_args()


class Stage:
    """HIP-event stopwatch on the current stream (our kernels are launched on torch's current stream).  Adjacent
    stages share ONE boundary event (an event record between two kernels costs a few microseconds of dispatch gap,
    two of them twice that)."""

    def __init__(self):
        self.t = {}
        self._open = {}
        self._last = None  # (event, was it a stop?)

    def _boundary(self, reuse):
        if reuse and self._last is not None:
            return self._last
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def start(self, name):
        # a start right after a stop (nothing launched in between) reuses the stop's event
        e = self._boundary(reuse=True)
        self._last = None
        self._open[name] = e

    def stop(self, name):
        e = self._boundary(reuse=False)
        self._last = e
        self.t.setdefault(name, []).append((self._open.pop(name), e))

    def launched(self):
        """Call after launching work that belongs to no stage: the next start gets its own event."""
        self._last = None

    def ms(self, name):
        return [a.elapsed_time(b) for a, b in self.t.get(name, [])]


def one_step(ops, obs, pred, K, max_iter, first_index, sw, km=None, timing=None, comm=None, reference_order=False):
    """The hot path once.  Returns (n_iter of the Lloyd loop).  `reference_order`: the k-means (farthest-first and Lloyd
    loop) in ATen's own fp32 summation orders (BatchKMeans(sums="reference-order"): the reference's labels bit for bit)
    instead of the exact partition-independent sums."""
    n = obs.shape[0]
    mode = ops.MODE_MOVING
    sw.start("fit")
    if km is None:
        U_obs, U_pred = ops.fit_descriptor(obs, pred, 6, mode, 0.0, 1)[:2]  # (et_fit_descriptor: Gram, reduction, one launch for both eigenproblems)
    else:
        from eigentrajectory_amd.dist import fit_descriptor_sharded
        U_obs, U_pred, _, _, _ = fit_descriptor_sharded(obs, pred, 6, mode, 0.0, 1, want_count=False, comm=comm)
    sw.stop("fit")
    sw.start("project")
    c_obs, c_pred, nrm, _ = ops.norm_project(obs, pred, U_obs, U_pred, None, None, mode, want_flag=False)
    sw.stop("project")
    sw.start("reconstruct")
    rec = ops.anchor_reconstruct(c_pred.view(6, n, 1), None, None, U_pred, None, mode, nrm=nrm)
    sw.stop("reconstruct")
    del rec, c_obs
    if km is None and reference_order:
        sw.start("kmeans_init")
        c0 = ops.kmeans_init_farthest_reference_order(c_pred, K, first_index)
        sw.stop("kmeans_init")
        sw.start("kmeans_lloyd")
        res = ops.kmeans_fit_reference_order(c_pred, c0, max_iter, 1e-4, trace=False)
        sw.stop("kmeans_lloyd")
    elif km is None:
        sw.start("kmeans_init")
        c0 = ops.kmeans_init_farthest(c_pred, K, first_index)
        sw.stop("kmeans_init")
        sw.start("kmeans_lloyd")
        # no per-iteration trace, like the reference (it only prints one when verbose); ET_BENCH_TRACE=1 records it
        res = ops.kmeans_fit(c_pred, c0, max_iter, 1e-4, timing=True, trace=os.environ.get("ET_BENCH_TRACE") == "1")
        sw.stop("kmeans_lloyd")
        # the first launch of a fit is the plain exact scan (kmeans_assign_kernel<6,4>, full accumulation); the
        # others are the filter kernel, the dominant kernel of the path, of which four launches out of every eight are timed
        timing.append((res["assign_ms"], res["assign_launches"], res["assign_iterations"]))
    else:
        skm = km(c_pred, K)
        sw.launched()
        sw.start("kmeans_init")
        c0 = skm.init_farthest(first_index)
        sw.stop("kmeans_init")
        sw.start("kmeans_lloyd")
        res = skm.fit(c0, max_iter, 1e-4)
        sw.stop("kmeans_lloyd")
    return res["n_iter"]


# the dominant kernel of the step: one launch per Lloyd iteration for shards above 32768 points (the chained kernel),
# ONE persistent launch for all iterations of a fit below that (csrc/et_kmeans.hip: km_persist_wanted; ET_OPT_KMEANS_LOOP
# forces a form).  Which one ran is read off the timing record (iterations per launch).
# Kernels are matched by PREFIX against the names rocprofv3 prints (the template argument list has grown over the rounds:
# <10, false> -> <10, false, false>); of several instantiations with the prefix the one with the most calls is the
# per-iteration launch.  Trace-less fits of shards >= 2^17 points iterate on a packed f16 copy of the points (14 B per
# point instead of 24): the same kernel, the packed body is a branch of it.
CHAIN_KERNEL = "et::kmeans_lloyd_chain_kernel<10, false"
PERSIST_KERNEL = "et::kmeans_lloyd_persist_kernel<10, false"


def packed_fits():
    """fits that iterated on the packed copy so far (a counter inside the library)"""
    import ctypes
    from eigentrajectory_amd import _lib as L
    fn = L.lib().et_internal_kmeans_packed_fits
    fn.restype = ctypes.c_longlong
    return int(fn())


def pmc_traffic(n, prefix):
    """HBM bytes per launch of the dominant kernel as measured by rocprofv3 PMC passes (FETCH_SIZE doubled as the gfx950
    guide prescribes, + WRITE_SIZE), from the NEWEST committed profile of the same workload size
    (profiles/*_pmc_hbm_traffic.json, made by tools/pmc_summary.py).  The kernel is matched by prefix; when the newest
    profile of this size does not hold it, that is an error to report (no silent fallback to an older profile).
    -> (bytes | None, source | None, kernel name as rocprof prints it | None, error | None)"""
    import glob
    tag = f"N={n:.0e}".replace("+0", "")
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.json")), reverse=True):
        rel = os.path.relpath(path, ROOT)
        try:
            js = json.load(open(path))
        except (OSError, ValueError) as exc:
            return None, rel, None, f"unreadable: {exc!r}"
        if tag not in js.get("note", "").replace("+0", ""):
            continue
        hits = {k: v for k, v in js.get("kernels", {}).items() if k.startswith(prefix)}
        if not hits:
            return None, rel, None, f"no kernel with prefix {prefix!r} among {sorted(js.get('kernels', {}))}"
        name = max(hits, key=lambda k: hits[k].get("calls", 0))
        return round(hits[name]["read_bytes_corrected"] + hits[name]["write_bytes"]), rel, name, None
    return None, None, None, f"no profiles/*_pmc_hbm_traffic.json for {tag}"


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _usable_cpus():
    """Logical CPUs this process may actually use: affinity mask and cgroup quota, not just os.cpu_count()."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()
            if q != "max":
                quota = float(q) / float(period)
    except (OSError, ValueError):
        pass
    return n, quota


def cpu_baseline(max_iter, budget_s, c_port_sample):
    """The reference's PyTorch-CPU path, timed on this box's host cores: oracle/torch_cpu_ref.py restates the ATen
    call sequence of descriptor.py / normalizer.py / kmeans.py (checked against the golden vectors in
    tests/test_torch_cpu_ref.py; the reference itself cannot travel to the GPU box), on the same synthetic
    workload.  Thread count: os.cpu_count() is tried first, then 64/32/16/8 -- intra-op parallelism over hundreds
    of threads is far slower than a few dozen for these small ATen ops (and a container's CPU quota can be below
    the visible core count), so the FASTEST setting is what gets reported, with the whole table next to it.
    N = 1e5: median of 3 full steps; N = 1e6: one step.  Every run carries a deadline: the Lloyd loop stops when
    it passes, and the iterations actually run are reported.  Reported, not the optimisation target.
    The scalar C oracle (one core) is kept as a second figure."""
    from eigentrajectory_amd.synth import synthetic_trajectories_np
    from oracle import et_oracle as eo
    from oracle import torch_cpu_ref as R
    n_cpu = os.cpu_count() or 1
    usable, quota = _usable_cpus()
    prev = torch.get_num_threads()
    try:
        obs, pred = (torch.from_numpy(a) for a in synthetic_trajectories_np(100_000, seed=0, min_disp=1e-3))
        tried = {}
        for th in sorted({n_cpu, usable, 64, 32, 16, 8}, reverse=True):
            if th > n_cpu or (quota is not None and th > 4 * quota):
                continue  # far beyond the container's CPU quota: measured once, 400x slower
            torch.set_num_threads(th)
            R.hot_path(obs[:5000], pred[:5000], first_index=100, max_iter=1)  # thread pool spin-up
            t0 = time.perf_counter()
            R.hot_path(obs[:30000], pred[:30000], first_index=15000, max_iter=2, deadline=t0 + 3.0)
            tried[th] = round(time.perf_counter() - t0, 3)
        threads = min(tried, key=tried.get)
        torch.set_num_threads(threads)
        sizes = {}
        runs = []
        for _ in range(3):
            st = {}
            R.hot_path(obs, pred, first_index=50_000, max_iter=max_iter, stages=st, deadline=time.perf_counter() + 6.0)
            runs.append(st)
        runs.sort(key=lambda d: d["total"])
        med = runs[1]
        sizes["1e5"] = dict(value=round(1e5 / med["total"], 1), seconds=round(med["total"], 3),
                            lloyd_iterations=med["lloyd_iterations"],
                            stages_s={k: round(v, 4) for k, v in med.items() if k not in ("total", "lloyd_iterations")})
        obs, pred = (torch.from_numpy(a) for a in synthetic_trajectories_np(1_000_000, seed=0, min_disp=1e-3))
        st = {}
        R.hot_path(obs, pred, first_index=500_000, max_iter=max_iter, stages=st,
                   deadline=time.perf_counter() + budget_s)
        sizes["1e6"] = dict(value=round(1e6 / st["total"], 1), seconds=round(st["total"], 3),
                            lloyd_iterations=st["lloyd_iterations"],
                            stages_s={k: round(v, 4) for k, v in st.items() if k not in ("total", "lloyd_iterations")})
        del obs, pred
    finally:
        torch.set_num_threads(prev)
    full = sizes["1e6"]["lloyd_iterations"] == max_iter
    head = sizes["1e6"] if full else sizes["1e5"]
    out = dict(value=head["value"], unit="trajectories/s", cores=threads, kind="port",
               impl="pytorch-restatement (oracle/torch_cpu_ref.py: torch.linalg.svd, U^T M, per-sample reconstruction "
                    "loop, BatchKMeans op sequence)",
               sample=(f"N={'1e6' if full else '1e5'} of the same synthetic workload, one full step "
                       f"(fit+project+reconstruct+k-means, {head['lloyd_iterations']} Lloyd iterations), "
                       f"{head['seconds']} s on {threads} threads (fastest of the settings tried); N=1e5 is the median of 3"),
               cpu_model=_cpu_model(), cpu_count=n_cpu, usable_cpus=usable, cgroup_cpu_quota=quota,
               seconds_by_threads=tried, torch=torch.__version__, sizes=sizes)
    # second figure: the scalar C restatement of the arithmetic (the parity oracle), one core
    eo.build()
    obs, pred = synthetic_trajectories_np(c_port_sample, seed=0, min_disp=1e-3)
    t0 = time.perf_counter()
    g_obs, g_pred, _ = eo.fit_gram(obs, pred, 1, 0.0, 1)
    U_obs, _ = eo.eigh_topk(g_obs, 6)
    U_pred, _ = eo.eigh_topk(g_pred, 6)
    _, c_pred, _, _ = eo.norm_project(obs, pred, U_obs, U_pred, None, None, 1)
    eo.anchor_reconstruct(c_pred[:, :, None], obs, None, None, U_pred, None, 1)
    c0, _ = eo.kmeans_init_farthest(c_pred, 20, c_port_sample // 2)
    res = eo.kmeans_fit(c_pred, c0, max_iter, 1e-4)
    dt = time.perf_counter() - t0
    out["c_port"] = dict(value=round(c_port_sample / dt, 1), unit="trajectories/s", cores=1,
                         sample=f"N={c_port_sample}, oracle/et_oracle.c single thread, {res['n_iter']} Lloyd iterations, "
                                f"{dt:.1f} s")
    return out


def _median_ms(fn, reps=5, warm=1):
    """Median HIP-event time of `fn` (launches on torch's current stream, like every op of this package)."""
    for _ in range(warm):
        fn()
    pairs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        pairs.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in pairs]))


def scene_latency(dev, U_obs, U_pred, n_peds=57, reps=2000):
    """The reference's real inference regime: one scene of N <= 57 pedestrians per wrapper call (model.py:58-125), 20
    samples, a predictor that returns zeros (so only the descriptor path and the hook plumbing are timed).  Wall
    time per call over `reps` back-to-back calls (the stream is drained once at the end): host + launch overhead,
    the kernels themselves take a few microseconds."""
    from eigentrajectory_amd import EigenTrajectory
    from eigentrajectory_amd.synth import synthetic_trajectories_torch
    from eigentrajectory_amd.utils import DotDict, default_hyper_params

    class Zero(torch.nn.Module):
        def forward(self, x):
            return torch.zeros((6, x.size(1), 20), device=x.device)

    hooks = DotDict(model_forward_pre_hook=lambda c, o, a=None: torch.cat([c, o], dim=0),
                    model_forward=lambda x, m: m(x), model_forward_post_hook=lambda y, a=None: y)
    model = EigenTrajectory(Zero(), hooks, default_hyper_params(static_dist=0.3)).to(dev)
    with torch.no_grad():
        for d_ in (model.ET_m_descriptor, model.ET_s_descriptor):
            d_.U_obs_trunc.copy_(U_obs)
            d_.U_pred_trunc.copy_(U_pred)
        model.ET_m_anchor.C_anchor.normal_()
        model.ET_s_anchor.C_anchor.normal_()
        obs, pred = synthetic_trajectories_torch(n_peds, dev, seed=5)
        res = {}
        # (..._replayed: the same call captured once per scene size in a HIP graph and replayed -- model.py)
        for name, fn in (("evaluate", lambda: model.evaluate(obs, pred)), ("forward", lambda: model(obs)),
                         ("evaluate_replayed", lambda: model.evaluate_replayed(obs, pred)),
                         ("forward_replayed", lambda: model.forward_replayed(obs))):
            for _ in range(50):
                fn()
            blocks = []  # (median of three blocks: single blocks of the replayed forms came out 2x slow now and then)
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps // 3):
                    fn()
                torch.cuda.synchronize()
                blocks.append((time.perf_counter() - t0) / (reps // 3) * 1e6)
            res[f"{name}_us_per_scene"] = round(sorted(blocks)[1], 2)
    # the TRAINING form of a wrapper call (utils/trainer.py:126-152): forward with the ground truth -> the three loss terms
    # -> backward into the predictor (here: into the refinement coefficients a one-parameter stub predictor emits, so
    # that what is timed is the descriptor path, its autograd and the loss arithmetic)
    class Bias(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.b = torch.nn.Parameter(torch.zeros((6, 1, 20)))

        def forward(self, x):
            return self.b.expand(6, x.size(1), 20) * 1.0

    tmodel = EigenTrajectory(Bias(), hooks, default_hyper_params(static_dist=0.3)).to(dev)
    tmodel.load_state_dict({**model.state_dict(), "baseline_model.b": tmodel.baseline_model.b.detach()})

    def train_step():
        with torch.enable_grad():  # (extra_stages runs under no_grad)
            tmodel.baseline_model.b.grad = None
            out = tmodel(obs, pred)
            (out["loss_eigentraj"] + out["loss_euclidean_ade"] + out["loss_euclidean_fde"]).backward()
    for _ in range(30):
        train_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps // 4):
        train_step()
    torch.cuda.synchronize()
    res["train_step_us_per_scene"] = round((time.perf_counter() - t0) / (reps // 4) * 1e6, 2)
    with torch.no_grad():
        # W1 (model.py:34-56) at the size of the reference's own fit sets (ETH train+val+flip: 70 316 trajectories):
        # two descriptor fits + two anchor clusterings in the reference's sklearn recipe (k-means++, n_init = 10)
        o, p = synthetic_trajectories_torch(70_316, dev, seed=6)
        model.calculate_parameters(o, p)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            model.calculate_parameters(o, p)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        res["calculate_parameters_ms_at_70316"] = round(min(ts) * 1e3, 2)
    res.update(n_peds=n_peds, samples=20, predictor="zero stub")
    return res


def step_roofline(n, ms, lloyd_iterations, K):
    """Algorithmic bytes of one whole step (SURVEY 8(d) per stage: fit 160 + project 208 + reconstruct 136 + (K - 1)
    farthest-first passes of 32 + 24 per Lloyd iteration + 8 of labels, per trajectory) against the HBM peak."""
    per = BYTES["fit"] + BYTES["project"] + BYTES["reconstruct"] + (K - 1) * BYTES["kmeans_init_step"] + \
        BYTES["kmeans_iter"] * lloyd_iterations + BYTES["labels"]
    gbs = per * n / ms / 1e6
    return dict(algorithmic_bytes_per_trajectory=round(per, 1), GBs=round(gbs, 1), frac_of_peak=round(gbs / HBM_PEAK_GBS, 4))


def extra_stages(ops, obs, pred, n, K, max_iter, first_index, dev):
    """Measurements beside the headline step (single GPU): the model form of the reconstruction (S = 20 samples,
    descriptor.py:162-176: (k,N,20) -> (20,N,12,2), 2416 B per trajectory) forward, backward and with the fused
    best-of-S metrics epilogue, and the whole step at N = 1e5 and 1e6 (BASELINE.json asks for all three sizes)."""
    from eigentrajectory_amd.synth import synthetic_trajectories_torch
    out, sizes = {}, {}
    mode = ops.MODE_MOVING
    with torch.no_grad():
        g_obs, g_pred, _ = ops.fit_gram(obs, pred, mode, 0.0, 1)
        (U_obs, _), (U_pred, _) = ops.eigh_topk_batch([g_obs, g_pred], 6)
        _, _, nrm, _, pose = ops.norm_project(obs, pred, U_obs, U_pred, None, None, mode, want_flag=False, want_pose=True)
        pose_split = ops.norm_project(obs, None, U_obs, None, U_obs, None, ops.MODE_SPLIT, 0.3, want_nrm=False, want_flag=False,
                                      want_obs=False, want_pose=True)[4]
        S = 20
        C20 = torch.randn((6, n, S), device=dev) * 0.1
        A = torch.randn((6, S), device=dev)
        rec = ops.anchor_reconstruct(C20, A, None, U_pred, None, mode, nrm=nrm)

        def stage(name, ms, nbytes):
            out[name] = dict(ms=round(ms, 4), bytes_per_traj=nbytes, GBs=round(nbytes * n / ms / 1e6, 1),
                             frac_of_peak=round(nbytes * n / ms / 1e6 / HBM_PEAK_GBS, 4))
        stage("reconstruct_S20_fwd", _median_ms(lambda: ops.anchor_reconstruct(C20, A, None, U_pred, None, mode, nrm=nrm)),
              480.0 + 16.0 + 1920.0)
        stage("reconstruct_S20_bwd", _median_ms(lambda: ops._reconstruct_bwd(rec, None, nrm, U_pred, None, mode, 0.0, 8)),
              1920.0 + 16.0 + 480.0)
        # the normaliser from the projection's optional pose record (5,N): origin, rotation x scale, 1 / scale -- 20 B per row
        # instead of nrm's 16, and no square root / reciprocal / selects per pass in the metric kernel
        stage("reconstruct_metrics_S20",
              _median_ms(lambda: ops.anchor_reconstruct_metrics(C20, pred, A, None, U_pred, None, mode, pose=pose)),
              480.0 + 20.0 + 96.0 + 8.0)
        stage("reconstruct_metrics_S20_from_nrm",
              _median_ms(lambda: ops.anchor_reconstruct_metrics(C20, pred, A, None, U_pred, None, mode, nrm=nrm)),
              480.0 + 16.0 + 96.0 + 8.0)
        # the wrapper's evaluate() form: descriptor chosen per row (model.py:46,73)
        stage("reconstruct_metrics_S20_per_row_descriptor",
              _median_ms(lambda: ops.anchor_reconstruct_metrics(C20, pred, A, A, U_pred, U_pred, ops.MODE_SPLIT, 0.3, pose=pose_split)),
              480.0 + 20.0 + 96.0 + 8.0)
        del rec, C20, pose, pose_split
        torch.cuda.empty_cache()
        out["scene_latency"] = scene_latency(dev, U_obs, U_pred)
        for tag, m in (("1e5", 100_000), ("1e6", 1_000_000)):
            o, p = synthetic_trajectories_torch(m, dev, seed=0, min_disp=1e-3)
            sw = Stage()
            for _ in range(3):
                one_step(ops, o, p, K, max_iter, first_index % m, sw, None, [])
            torch.cuda.synchronize()
            steps = 10
            t0 = time.perf_counter()
            its = [one_step(ops, o, p, K, max_iter, first_index % m, sw, None, []) for _ in range(steps)]
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            sizes[tag] = dict(value=round(m / dt, 1), unit="trajectories/s", ms_per_step=round(dt * 1e3, 4),
                              lloyd_iterations=float(np.mean(its)), **step_roofline(m, dt * 1e3, float(np.mean(its)), K))
    return out, sizes


def reference_order_lloyd(ops, dev, n_main, K, max_iter, first_index):
    """BatchKMeans(sums="reference-order") beside the default on the SAME k-means inputs (the projection of the synthetic
    trajectories) at N = 1e5 / 1e6 / n_main: the Lloyd loop in ATen's own summation orders (kmeans.py:180-182, 73-74,
    45-51; csrc/et_kmeans_reforder.hip) -- the path that reproduces the imported reference's whole runs bit for bit
    (tests/golden/g7c, g7d) -- against the exact-sum loop the headline step times.  Both loops between HIP events on the
    launch stream, per iteration; `labels_equal`: whether the two fits end with the same labels on this input (they
    differ by the ~1e-7 summation noise Lloyd iterations amplify: DESIGN 4)."""
    from eigentrajectory_amd.synth import synthetic_trajectories_torch
    out = {}
    for m in sorted({100_000, 1_000_000, int(n_main)}):
        with torch.no_grad():
            o, p = synthetic_trajectories_torch(m, dev, seed=0, min_disp=1e-3)
            g_obs, g_pred, _ = ops.fit_gram(o, p, ops.MODE_MOVING, 0.0, 1)
            (U_obs, _), (U_pred, _) = ops.eigh_topk_batch([g_obs, g_pred], 6)
            _, x, _, _ = ops.norm_project(o, p, U_obs, U_pred, None, None, ops.MODE_MOVING, want_flag=False, want_nrm=False)
            del o, p
            c0 = ops.kmeans_init_farthest(x, K, first_index % m)
            inits = {}
            for name, fn in (("reference_order", ops.kmeans_init_farthest_reference_order), ("exact_sum", ops.kmeans_init_farthest)):
                ts = []
                for rep in range(3):
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record()
                    c_init = fn(x, K, first_index % m)
                    ev1.record()
                    torch.cuda.synchronize()
                    if rep:
                        ts.append(ev0.elapsed_time(ev1))
                inits[name] = (min(ts), c_init)
            best = {}
            for name in ("reference_order", "exact_sum"):
                runs = []
                for rep in range(4):  # (the first one warms up)
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record()
                    r = (ops.kmeans_fit_reference_order(x, c0, max_iter, 1e-4, trace=False) if name == "reference_order"
                         else ops.kmeans_fit(x, c0, max_iter, 1e-4, trace=False))
                    ev1.record()
                    torch.cuda.synchronize()
                    if rep:
                        runs.append(ev0.elapsed_time(ev1))
                best[name] = (min(runs), r)
            ms_r, r_ref = best["reference_order"]
            ms_e, r_ex = best["exact_sum"]
            tag = f"{m:.0e}".replace("+0", "")
            per_r, per_e = ms_r / max(r_ref["n_iter"], 1) * 1e3, ms_e / max(r_ex["n_iter"], 1) * 1e3
            out[tag] = dict(fit_ms=round(ms_r, 4), iterations=r_ref["n_iter"], us_per_iteration=round(per_r, 2),
                            exact_sum_fit_ms=round(ms_e, 4), exact_sum_iterations=r_ex["n_iter"],
                            exact_sum_us_per_iteration=round(per_e, 2), ratio=round(per_r / per_e, 3),
                            algorithmic_GBs=round(BYTES["kmeans_iter"] * m / per_r / 1e3, 1),
                            frac_of_peak=round(BYTES["kmeans_iter"] * m / per_r / 1e3 / HBM_PEAK_GBS, 4),
                            labels_equal=bool(r_ref["n_iter"] == r_ex["n_iter"] and torch.equal(r_ref["labels"], r_ex["labels"])),
                            farthest_first_ms=round(inits["reference_order"][0], 4), exact_sum_farthest_first_ms=round(inits["exact_sum"][0], 4),
                            same_initial_centroids=bool(torch.equal(inits["reference_order"][1], inits["exact_sum"][1])))
            del x
            torch.cuda.empty_cache()
    out["note"] = ("whole fits (<= max_iter Lloyd iterations, incl. the one-off permuted copy / packed copy of the points) between "
                   "events; reference_order reads 24 B of coordinates per point and iteration once (assignment and cascade levels in "
                   "one kernel) + 1.9 KB of partial sums per 1024-4096 points")
    return out


def lloyd_beyond_cache(ops, dev, K, max_iter, first_index, n_big=40_000_000):
    """The dominant kernel on a shard whose working set does NOT fit the 256 MB Infinity Cache: at N = 1e7 an iteration
    streams 140 MB of packed rows + 10 MB of labels -- cache resident, so the headline `roofline` is an algorithmic rate,
    not HBM traffic.  N = 4e7: 560 + 40 MB per iteration, every iteration from HBM.  One k-means fit (farthest-first +
    Lloyd) on the projection of 4 x 1e7 synthetic trajectories; same timing fields as the headline."""
    from eigentrajectory_amd.synth import synthetic_trajectories_torch
    chunk = 10_000_000
    with torch.no_grad():
        c = torch.empty((6, n_big), device=dev)
        U = None
        for i in range(n_big // chunk):
            o, p = synthetic_trajectories_torch(chunk, dev, seed=100 + i, min_disp=1e-3)
            if U is None:
                g_obs, g_pred, _ = ops.fit_gram(o, p, ops.MODE_MOVING, 0.0, 1)
                (U_obs, _), (U_pred, _) = ops.eigh_topk_batch([g_obs, g_pred], 6)
                U = (U_obs, U_pred)
            _, cp, _, _ = ops.norm_project(o, p, U[0], U[1], None, None, ops.MODE_MOVING, want_flag=False, want_nrm=False)
            c[:, i * chunk:(i + 1) * chunk] = cp
            del o, p, cp
        torch.cuda.empty_cache()
        c0 = ops.kmeans_init_farthest(c, K, first_index)
        ops.kmeans_fit(c, c0, 3, 1e-4, trace=False)  # warm-up
        res = ops.kmeans_fit(c, c0, max_iter, 1e-4, timing=True, trace=False)
        del c
        torch.cuda.empty_cache()
    avg_ms = res["assign_ms"] / max(res["assign_launches"], 1)
    return dict(n=n_big, iterations=res["n_iter"], avg_launch_ms=round(avg_ms, 5),
                algorithmic_GBs=round(BYTES["kmeans_iter"] * n_big / avg_ms / 1e6, 1),
                algorithmic_frac=round(BYTES["kmeans_iter"] * n_big / avg_ms / 1e6 / HBM_PEAK_GBS, 4),
                moved_bytes_per_point=15.0,
                moved_GBs=round(15.0 * n_big / avg_ms / 1e6, 1), moved_frac=round(15.0 * n_big / avg_ms / 1e6 / HBM_PEAK_GBS, 4),
                note="packed f16 rows (14 B) + labels (1 B) per point and iteration = 600 MB > the 256 MB Infinity Cache: "
                     "`moved_*` is HBM traffic here (at N = 1e7 the 150 MB of an iteration stay cache resident)")


def scaling_model(ops, obs, pred, K, max_iter, first_index, dev, single_ms, n_it):
    """What the first multi-rank run should show (none could be measured: the pool has one GPU per box).  Measured here:
    the sharded path on ONE rank through a real RCCL communicator (the library enqueues ncclAllReduce / ncclAllGather
    between its launches; nobody to add).  Modelled: per exchange the difference between an assumed P-rank latency of
    RCCL's small-message collectives over xGMI and the one-rank latency measured in this run.  Weak scaling (every rank
    its own N rows): the step time of P ranks = the one-rank sharded step + exchanges x that difference."""
    import torch.distributed as tdist
    from eigentrajectory_amd.dist import Communicator, ShardedKMeans
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    out = dict(measured_single_gpu_ms=round(single_ms, 3))
    try:
        tdist.init_process_group("nccl", device_id=dev)
        comm = Communicator(dev)
        km = (lambda x, k: ShardedKMeans(x, k, comm=comm))
        sw = Stage()
        for _ in range(2):
            one_step(ops, obs, pred, K, max_iter, first_index, sw, km, [], comm)
        torch.cuda.synchronize()
        sw = Stage()
        t0 = time.perf_counter()
        for _ in range(5):
            one_step(ops, obs, pred, K, max_iter, first_index, sw, km, [], comm)
        torch.cuda.synchronize()
        w1 = (time.perf_counter() - t0) / 5 * 1e3
        lloyd_w1 = float(np.mean(sw.ms("kmeans_lloyd")))
        comm.close()
        tdist.destroy_process_group()
    except Exception as exc:  # noqa: BLE001 -- the model is an extra: never lose the headline over it
        out["error"] = repr(exc)
        return out
    assumed_us = {1: 0.0, 2: 8.0, 4: 12.0, 8: 16.0}  # extra latency of a <= 1.1 KB all-reduce / all-gather over P ranks vs one rank
    exchanges = n_it + (K - 1) + 2  # one all-reduce per Lloyd iteration, one all-gather per farthest-first step, Gram + scale scan
    pred_ms = {p_: round(w1 + exchanges * us * 1e-3, 3) for p_, us in assumed_us.items()}
    out.update(measured_sharded_world1_ms=round(w1, 3), measured_sharded_world1_lloyd_ms=round(lloyd_w1, 3),
               exchanges_per_step=exchanges, assumed_extra_latency_us_per_exchange=assumed_us,
               predicted_ms_per_step=pred_ms,
               predicted_weak_scaling_efficiency={p_: round(single_ms / v, 3) for p_, v in pred_ms.items()},
               note="efficiency = single-GPU step / P-rank step at the same rows per rank; the assumed latencies are "
                    "the only unmeasured input")
    return out


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (no WORLD_SIZE in the environment): start the N
    ranks ourselves -- one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1 -- and hand their
    output through.  Returns the exit code."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs, this machine has {n_dev}", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    if torch.cuda.device_count() <= local_rank:
        print(f"bench.py: rank {rank} needs GPU {local_rank}, this machine has {torch.cuda.device_count()}", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dist = os.environ.get("ET_BENCH_FORCE_DIST") == "1"  # exercise the RCCL path on one GPU (testing aid)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    from eigentrajectory_amd import ops
    from eigentrajectory_amd.synth import synthetic_trajectories_torch
    n = int(args.n)
    K = 20
    obs, pred = synthetic_trajectories_torch(n, dev, seed=rank, min_disp=1e-3)
    km = comm = None
    native_ranks = 0
    dist_path = "single"
    if world > 1 or force_dist:
        from eigentrajectory_amd.dist import Communicator, ShardedKMeans
        # default: the native sharded entry points (the library enqueues its RCCL collectives on the stream, no Python
        # inside the Lloyd loop); ET_BENCH_DIST=torch drives the step API with torch.distributed collectives instead
        if os.environ.get("ET_BENCH_DIST", "native") == "native":
            try:
                comm = Communicator(dev)
                native_ranks = comm.info()[0]  # ncclCommCount of the communicator the library enqueues its collectives on
                dist_path = "native (et_kmeans_fit_sharded, RCCL ranks seen by the library: %d)" % native_ranks
            except Exception as exc:  # noqa: BLE001 -- fall back loudly, keep the run alive
                print(f"[bench] native RCCL communicator unavailable ({exc!r}); using torch.distributed collectives",
                      file=sys.stderr, flush=True)
                comm = None
            if world > 1:  # every rank must take the same path: one failure sends all of them to the step API
                ok = torch.tensor([0 if comm is None else 1], device=dev, dtype=torch.int32)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0 and comm is not None:
                    comm.close()
                    comm = None
        if comm is None:
            dist_path = "torch.distributed step API"
        km = (lambda x, k: ShardedKMeans(x, k, comm=comm))
    first_index = 12345

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    warm = Stage()
    for _ in range(args.warmup):
        one_step(ops, obs, pred, K, args.max_iter, first_index, warm, km, [], comm)
    sw = Stage()
    timing = []
    packed_before = packed_fits()
    barrier()
    t0 = time.perf_counter()
    iters, step_ends = [], []
    for _ in range(args.steps):
        iters.append(one_step(ops, obs, pred, K, args.max_iter, first_index, sw, km, timing, comm))
        step_ends.append(time.perf_counter())  # (a step ends synchronised: the k-means fit hands its iteration count to the host)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank must have run the same Lloyd loop (identical integers in, identical convergence flag out): a
        # disagreement means the ranks' collectives did not pair up -- say so loudly instead of printing a number
        mine = torch.tensor(iters, device=dev, dtype=torch.int64)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        if any(not torch.equal(e, every[0]) for e in every):
            if rank == 0:
                print("bench.py: the ranks disagree on the Lloyd iteration counts per step: " +
                      "; ".join(f"rank {r}: {e.tolist()}" for r, e in enumerate(every)), file=sys.stderr, flush=True)
            dist.destroy_process_group()
            sys.exit(3)
    step_ms = np.diff(np.asarray([t0] + step_ends)) * 1e3

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        total_traj = float(n) * world
        stages = {}
        for name, per in (("fit", BYTES["fit"]), ("project", BYTES["project"]), ("reconstruct", BYTES["reconstruct"])):
            ms = float(np.mean(sw.ms(name)))
            stages[name] = dict(ms=round(ms, 4), GBs=round(per * n / ms / 1e6, 1), frac_of_peak=round(per * n / ms / 1e6 / HBM_PEAK_GBS, 4))
        # no GB/s here: the farthest-first steps skip most coordinate reads (triangle-inequality bound), so neither
        # the un-skipped algorithmic bytes nor a fixed fraction of them describes what the kernel moves
        stages["kmeans_init"] = dict(ms=round(float(np.mean(sw.ms("kmeans_init"))), 4), steps=K - 1)
        ms = float(np.mean(sw.ms("kmeans_lloyd")))
        n_it = float(np.mean(iters))
        stages["kmeans_lloyd"] = dict(ms=round(ms, 4), iterations=n_it,
                                      GBs=round((BYTES["kmeans_iter"] * n_it + BYTES["labels"]) * n / ms / 1e6, 1))
        pr = stages["project"]["ms"] + stages["reconstruct"]["ms"]
        stages["project+reconstruct"] = dict(ms=round(pr, 4), GBs=round(344.0 * n / pr / 1e6, 1),
                                             frac_of_peak=round(344.0 * n / pr / 1e6 / HBM_PEAK_GBS, 4))
        # dominant kernel by time: the Lloyd kernel (kmeans_lloyd_chain_kernel<10,false>: one launch per iteration), timed
        # with HIP events recorded on the launch stream around a sample of the launches inside the timed steps
        # (et_kmeans_fit: the first launch, and of the others runs of four out of every eight between one pair of events)
        if timing and sum(c for _, c, _ in timing) > 0:
            launches = sum(c for _, c, _ in timing)
            avg_ms = sum(m for m, _, _ in timing) / launches
            its_per_launch = sum(i for _, _, i in timing) / launches  # Lloyd iterations one launch covers (1 when chained)
        else:  # sharded runs: the library enqueues launches and collectives itself; fall back to the loop average
            avg_ms = stages["kmeans_lloyd"]["ms"] / max(n_it, 1.0)
            its_per_launch = 1.0
        # algorithmic bytes of ONE launch of the dominant kernel: 24 B of coordinates per point and Lloyd iteration
        #  x the iterations that launch runs
        alg_bytes = BYTES["kmeans_iter"] * n * its_per_launch
        achieved = alg_bytes / avg_ms / 1e6
        dominant = PERSIST_KERNEL if its_per_launch > 1.5 else CHAIN_KERNEL
        traffic, traffic_source, rocprof_name, traffic_error = pmc_traffic(n, dominant)
        packed = its_per_launch <= 1.5 and packed_fits() > packed_before
        # `traffic` is NOT measured in this run: it is the PMC figure of the committed rocprofv3 passes of the same
        # workload (`traffic_source`); everything else on the line is measured live.  `bound`: the roof the fraction is
        # priced against (24 algorithmic bytes per point and iteration against the HBM peak); `limited_by`:
        # what the counters say holds the kernel (profiles/*_sq_breakdown.txt, DESIGN 3: at N = 1e7 the 150 MB an
        # iteration touches stay Infinity-Cache resident and the pass is bound by vector-instruction issue)
        roofline = dict(bound="hbm", kernel=(rocprof_name or dominant + ", ...>").replace("et::", ""),
                        limited_by=("vector-issue (packed rows Infinity-Cache resident)" if packed and n <= 12_000_000
                                    else "hbm"),
                        achieved=round(achieved, 1), peak=HBM_PEAK_GBS,
                        unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_source,
                        avg_launch_ms=round(avg_ms, 5), lloyd_iterations_per_launch=round(its_per_launch, 2),
                        algorithmic_bytes_per_launch=alg_bytes,
                        points_read_as=("packed f16 rows, 14 B per point" if packed else "fp32 rows, 24 B per point"))
        if traffic_error:
            roofline["traffic_error"] = traffic_error
        out = dict(metric="trajectories/sec fit+project+reconstruct+kmeans", value=total_traj / (elapsed / args.steps),
                   unit="trajectories/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(ms_per_step, 3),
                   ms_per_step_spread=dict(min=round(float(step_ms.min()), 3), median=round(float(np.median(step_ms)), 3),
                                           max=round(float(step_ms.max()), 3), note="rank 0, per step of the timed region"),
                   lloyd_iterations_per_step=[int(i) for i in iters],
                   higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype="f32", data="synthetic",
                   config=dict(workload=f"synthetic N={n:.0e} trajectories per GPU (obs 8 / pred 12 steps), k=6, "
                                        f"fit + project(obs+pred) + reconstruct(S=1) + k-means(K=20, farthest-first, "
                                        f"max_iter={args.max_iter}, tol=1e-4)",
                               n_per_gpu=n, k=6, num_clusters=K, parallelism=f"shard{world}",
                               rccl_ranks=(native_ranks if comm is not None else
                                           (dist.get_world_size() if dist.is_initialized() else 0)), dist_path=dist_path),
                   roofline=roofline, stages=stages)
        if world == 1 and not force_dist and not args.no_extras:
            # the same step with the k-means in the REFERENCE's summation orders (the mode that ends with the imported
            # reference's labels bit for bit: tests/golden g7, g7c, g7d) -- `value` times the exact-sum default
            sw_ro = Stage()
            one_step(ops, obs, pred, K, args.max_iter, first_index, sw_ro, reference_order=True)
            torch.cuda.synchronize()
            sw_ro = Stage()
            t_ro = time.perf_counter()
            its_ro = [one_step(ops, obs, pred, K, args.max_iter, first_index, sw_ro, reference_order=True) for _ in range(3)]
            torch.cuda.synchronize()
            ms_ro = (time.perf_counter() - t_ro) / 3 * 1e3
            out["value_reference_order"] = round(n / ms_ro * 1e3, 1)
            out["reference_order_step"] = dict(
                ms_per_step=round(ms_ro, 3), lloyd_iterations_per_step=[int(i) for i in its_ro],
                stages_ms={k: round(float(np.mean(sw_ro.ms(k))), 4) for k in ("fit", "project", "reconstruct", "kmeans_init", "kmeans_lloyd")},
                note="fit + project + reconstruct as in `value`; farthest-first and the Lloyd loop through "
                     "et_kmeans_*_reforder (ATen's fp32 cascade order, kmeans.py:180-182)")
            more, sizes = extra_stages(ops, obs, pred, n, K, args.max_iter, first_index, dev)
            stages.update(more)
            stages["kmeans_lloyd_reference_order"] = reference_order_lloyd(ops, dev, n, K, args.max_iter, first_index)
            if n == 10_000_000:
                roofline["beyond_infinity_cache"] = lloyd_beyond_cache(ops, dev, K, args.max_iter, first_index)
            out["scaling_model"] = scaling_model(ops, obs, pred, K, args.max_iter, first_index, dev, ms_per_step,
                                                 float(np.mean(iters)))
            sizes[f"{n:.0e}".replace("+0", "")] = dict(value=round(out["value"], 1), unit="trajectories/s",
                                                       ms_per_step=out["ms_per_step"], lloyd_iterations=float(np.mean(iters)),
                                                       **step_roofline(n, ms_per_step, float(np.mean(iters)), K))
            out["sizes"] = sizes
        if world == 1 and not args.no_cpu_baseline:
            del obs, pred
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(args.max_iter, args.cpu_budget, args.c_port_sample)
    if comm is not None:
        comm.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        # the ONE JSON line goes out last: anything the runtime libraries still hold in C stdio buffers (RCCL prints
        # a version banner) is flushed first
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
