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
    ap.add_argument("--cpu-sample", type=int, default=600_000,
                    help="trajectories of the CPU-baseline sample (default: ~10 s of single-core work)")
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


def one_step(ops, obs, pred, K, max_iter, first_index, sw, km=None, timing=None):
    """The hot path once.  Returns (n_iter of the Lloyd loop)."""
    n = obs.shape[0]
    mode = ops.MODE_MOVING
    sw.start("fit")
    if km is None:
        g_obs, g_pred, _ = ops.fit_gram(obs, pred, mode, 0.0, 1)
        (U_obs, _), (U_pred, _) = ops.eigh_topk_batch([g_obs, g_pred], 6)
    else:
        from eigentrajectory_amd.dist import fit_descriptor_sharded
        U_obs, U_pred, _, _, _ = fit_descriptor_sharded(obs, pred, 6, mode, 0.0, 1, want_count=False)
    sw.stop("fit")
    sw.start("project")
    c_obs, c_pred, nrm, _ = ops.norm_project(obs, pred, U_obs, U_pred, None, None, mode, want_flag=False)
    sw.stop("project")
    sw.start("reconstruct")
    rec = ops.anchor_reconstruct(c_pred.view(6, n, 1), None, None, U_pred, None, mode, nrm=nrm)
    sw.stop("reconstruct")
    del rec, c_obs
    if km is None:
        sw.start("kmeans_init")
        c0 = ops.kmeans_init_farthest(c_pred, K, first_index)
        sw.stop("kmeans_init")
        sw.start("kmeans_lloyd")
        res = ops.kmeans_fit(c_pred, c0, max_iter, 1e-4, timing=True)
        sw.stop("kmeans_lloyd")
        # the first launch of a fit is the plain exact scan (kmeans_assign_kernel<6,4>, full accumulation); the
        # others are the filter kernel, the dominant kernel of the path, of which every 8th launch is timed
        timing.append((res["assign_ms"], res["assign_launches"]))
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


DOMINANT_KERNEL = "et::kmeans_assign_filter_kernel<10>"  # Lloyd iterations >= 1 (99 of 100 launches per step)


def pmc_traffic(n):
    """HBM bytes per launch of the dominant kernel as measured by rocprofv3 PMC passes (FETCH_SIZE doubled
    as the gfx950 guide prescribes, + WRITE_SIZE); taken from the committed profile of the same workload
    size (profiles/*_pmc_hbm_traffic.json, made by tools/pmc_summary.py), else null."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.json")), reverse=True):
        try:
            js = json.load(open(path))
            if f"N={n:.0e}".replace("+0", "") not in js.get("note", "").replace("+0", ""):
                continue
            k = js["kernels"][DOMINANT_KERNEL]
            return round(k["read_bytes_corrected"] + k["write_bytes"])
        except Exception:
            continue
    return None


def cpu_baseline(sample_n, max_iter):
    """The CPU oracle (scalar C restatement of the reference's algorithm, one core) on a bounded sample
    of the same workload; reported, not the optimisation target."""
    from eigentrajectory_amd.synth import synthetic_trajectories_np
    from oracle import et_oracle as eo
    eo.build()
    obs, pred = synthetic_trajectories_np(sample_n, seed=0, min_disp=1e-3)
    t0 = time.perf_counter()
    g_obs, g_pred, _ = eo.fit_gram(obs, pred, 1, 0.0, 1)
    U_obs, _ = eo.eigh_topk(g_obs, 6)
    U_pred, _ = eo.eigh_topk(g_pred, 6)
    _, c_pred, _, _ = eo.norm_project(obs, pred, U_obs, U_pred, None, None, 1)
    eo.anchor_reconstruct(c_pred[:, :, None], obs, None, None, U_pred, None, 1)
    c0, _ = eo.kmeans_init_farthest(c_pred, 20, sample_n // 2)
    res = eo.kmeans_fit(c_pred, c0, max_iter, 1e-4)
    dt = time.perf_counter() - t0
    return dict(value=sample_n / dt, unit="trajectories/s", cores=1, kind="port",
                sample=f"N={sample_n} of the same synthetic workload, oracle/et_oracle.c single thread, "
                       f"{res['n_iter']} Lloyd iterations, {dt:.1f} s")


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dist = os.environ.get("ET_BENCH_FORCE_DIST") == "1"  # exercise the RCCL path on one GPU (testing aid)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
    assert world == args.gpus or (args.gpus == 1 and world == 1), f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from eigentrajectory_amd import ops
    from eigentrajectory_amd.synth import synthetic_trajectories_torch
    n = int(args.n)
    K = 20
    obs, pred = synthetic_trajectories_torch(n, dev, seed=rank, min_disp=1e-3)
    km = None
    if world > 1 or force_dist:
        from eigentrajectory_amd.dist import ShardedKMeans
        km = ShardedKMeans
    first_index = 12345

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    warm = Stage()
    for _ in range(args.warmup):
        one_step(ops, obs, pred, K, args.max_iter, first_index, warm, km, [])
    sw = Stage()
    timing = []
    barrier()
    t0 = time.perf_counter()
    iters = []
    for _ in range(args.steps):
        iters.append(one_step(ops, obs, pred, K, args.max_iter, first_index, sw, km, timing))
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        total_traj = float(n) * world
        stages = {}
        for name, per in (("fit", BYTES["fit"]), ("project", BYTES["project"]), ("reconstruct", BYTES["reconstruct"])):
            ms = float(np.mean(sw.ms(name)))
            stages[name] = dict(ms=round(ms, 4), GBs=round(per * n / ms / 1e6, 1), frac_of_peak=round(per * n / ms / 1e6 / HBM_PEAK_GBS, 4))
        ms = float(np.mean(sw.ms("kmeans_init")))
        stages["kmeans_init"] = dict(ms=round(ms, 4), GBs=round(BYTES["kmeans_init_step"] * (K - 1) * n / ms / 1e6, 1))
        ms = float(np.mean(sw.ms("kmeans_lloyd")))
        n_it = float(np.mean(iters))
        stages["kmeans_lloyd"] = dict(ms=round(ms, 4), iterations=n_it,
                                      GBs=round((BYTES["kmeans_iter"] * n_it + BYTES["labels"]) * n / ms / 1e6, 1))
        pr = stages["project"]["ms"] + stages["reconstruct"]["ms"]
        stages["project+reconstruct"] = dict(ms=round(pr, 4), GBs=round(344.0 * n / pr / 1e6, 1),
                                             frac_of_peak=round(344.0 * n / pr / 1e6 / HBM_PEAK_GBS, 4))
        # dominant kernel by time: the Lloyd assign kernel of iterations >= 1 (kmeans_assign_filter_kernel<10>),
        # timed with HIP events recorded on the launch stream around every launch inside the timed steps
        # (et_kmeans_fit)
        if timing and sum(c for _, c in timing) > 0:
            avg_ms = sum(m for m, _ in timing) / sum(c for _, c in timing)
        else:  # sharded runs drive the step API from Python; fall back to the loop average
            avg_ms = stages["kmeans_lloyd"]["ms"] / max(n_it, 1.0)
        achieved = BYTES["kmeans_iter"] * n / avg_ms / 1e6
        roofline = dict(bound="hbm", kernel=DOMINANT_KERNEL.replace("et::", ""), achieved=round(achieved, 1), peak=HBM_PEAK_GBS,
                        unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4), traffic=pmc_traffic(n),
                        avg_launch_ms=round(avg_ms, 5), algorithmic_bytes_per_launch=BYTES["kmeans_iter"] * n)
        out = dict(metric="trajectories/sec fit+project+reconstruct+kmeans", value=total_traj / (elapsed / args.steps),
                   unit="trajectories/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(ms_per_step, 3), higher_is_better=True, scaling="weak", vs_baseline=None,
                   dtype="f32", data="synthetic",
                   config=dict(workload=f"synthetic N={n:.0e} trajectories per GPU (obs 8 / pred 12 steps), k=6, "
                                        f"fit + project(obs+pred) + reconstruct(S=1) + k-means(K=20, farthest-first, "
                                        f"max_iter={args.max_iter}, tol=1e-4)",
                               n_per_gpu=n, k=6, num_clusters=K, parallelism=f"shard{world}"),
                   roofline=roofline, stages=stages)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.max_iter)
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
