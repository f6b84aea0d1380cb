"""EigenTrajectory -- the reference's wrapper (EigenTrajectory/model.py) on fused HIP kernels.

Reference: EigenTrajectory/model.py:7-125.  Drop-in: same constructor, the same
sub-module names (``ET_m_descriptor``, ``ET_s_descriptor``, ``ET_m_anchor``,
``ET_s_anchor``, ``baseline_model``) and therefore the same ``state_dict`` keys, the
same three-hook predictor protocol (model.py:93-95) and the same output dict.

What changes is how the descriptor path runs.  The reference physically splits
the batch into moving / static pedestrians with boolean masks (each mask is a
nonzero + gather + host sync: model.py:73-77, 82-83, 86-90, 98-99, 104-105,
110-115) and runs two descriptors.  Here every kernel takes the whole scene and
routes each row to its descriptor by the same test (model.py:73), so a forward
is: 1 projection kernel -> predictor -> 1 reconstruction kernel, no host syncs.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .anchor import ETAnchor
from .descriptor import ETDescriptor


class _SceneLosses(torch.autograd.Function):
    """Anchor refinement + reconstruction + the three loss terms of model.py:119-123 for one scene in ONE launch, their
    gradient w.r.t. the predictor's output in one more (csrc/et_train.hip).  Differentiable w.r.t. C only, like the
    reference (anchors, U and the normaliser state are detached there too: anchor.py:87, descriptor.py:72,87)."""

    @staticmethod
    def forward(ctx, Cc, model, nrm, C_gt, gt, t_obs):
        k, n, s = Cc.shape
        t_pred = gt.shape[1]
        dev = Cc.device
        recon = torch.empty((s, n, t_pred, 2), device=dev)
        small = torch.empty((3 * n + 3,), device=dev)           # best (3,N) | losses (3)
        arg = torch.empty((3, n), device=dev, dtype=torch.int32)
        ptrs = (nrm.data_ptr(), model.ET_m_anchor.C_anchor.data_ptr(), model.ET_s_anchor.C_anchor.data_ptr(),
                model.ET_m_descriptor.U_pred_trunc.data_ptr(), model.ET_s_descriptor.U_pred_trunc.data_ptr())
        rc = ops.L.fast("et_wrapper_losses_fwd")(
            Cc.data_ptr(), n, s, k, t_pred, *ptrs, ops.MODE_SPLIT, model.static_dist, C_gt.data_ptr(), gt.data_ptr(),
            recon.data_ptr(), small.data_ptr(), arg.data_ptr(), small.data_ptr() + 12 * n, ops.L.raw_stream(dev.index))
        if rc:
            ops.L.check(rc, "et_wrapper_losses_fwd")
        ctx.save_for_backward(Cc, nrm, C_gt, gt, recon, arg)
        # outputs nobody differentiates arrive as None in backward (autograd would otherwise hand over zero tensors: an
        # (S,N,T,2) fill + a second backward launch + an add in every training step)
        ctx.set_materialize_grads(False)
        ctx.ptrs, ctx.static_dist, ctx.t_obs = ptrs, model.static_dist, t_obs
        ctx.U = (model.ET_m_descriptor.U_pred_trunc.detach(), model.ET_s_descriptor.U_pred_trunc.detach())
        return recon, small[3 * n], small[3 * n + 1], small[3 * n + 2]

    @staticmethod
    def backward(ctx, g_recon, g_e, g_ade, g_fde):
        Cc, nrm, C_gt, gt, recon, arg = ctx.saved_tensors
        k, n, s = Cc.shape
        dC = torch.empty_like(Cc)

        def scalar(g):  # the upstream gradient of one loss term: a 0-dim fp32 tensor on the device, or nothing
            if g is None:
                return None
            if g.dtype != torch.float32 or not g.is_cuda:
                g = g.to(device=Cc.device, dtype=torch.float32)
            return g
        gs = [scalar(g) for g in (g_e, g_ade, g_fde)]
        rc = ops.L.fast("et_wrapper_losses_bwd")(
            *(None if g is None else g.data_ptr() for g in gs), Cc.data_ptr(), n, s, k, gt.shape[1], *ctx.ptrs, ops.MODE_SPLIT,
            ctx.static_dist, C_gt.data_ptr(), gt.data_ptr(), recon.data_ptr(), arg.data_ptr(), dC.data_ptr(),
            ops.L.raw_stream(Cc.device.index))
        if rc:
            ops.L.check(rc, "et_wrapper_losses_bwd")
        if g_recon is not None:  # somebody also differentiates through recon_traj itself
            dC = dC + ops._reconstruct_bwd(g_recon.contiguous().float(), None, nrm, ctx.U[0], ctx.U[1], ops.MODE_SPLIT,
                                           ctx.static_dist, ctx.t_obs)
        return dC, None, None, None, None, None


_SIDE = {}


def _side_lane(dev):
    """(one-thread executor, side stream) of a device for work that runs next to the caller's stream."""
    key = (dev.type, dev.index)
    if key not in _SIDE:
        from concurrent.futures import ThreadPoolExecutor
        _SIDE[key] = (ThreadPoolExecutor(max_workers=1, thread_name_prefix="et-side"), torch.cuda.Stream(device=dev))
    return _SIDE[key]


class EigenTrajectory(nn.Module):
    r"""Wrapper that runs any trajectory predictor in the ET coefficient space (model.py:9-32 of the reference).

    ``baseline_model``: the predictor network; ``hook_func``: its three bridge functions (``model_forward_pre_hook``,
    ``model_forward``, ``model_forward_post_hook``); ``hyper_params``: DotDict with ``obs_len, pred_len, obs_svd,
    pred_svd, k, num_samples, traj_dim, static_dist``.  Sub-module and attribute names are the reference's, so its
    checkpoints load unchanged.
    """

    def __init__(self, baseline_model, hook_func, hyper_params):
        super().__init__()

        self.baseline_model = baseline_model
        self.hook_func = hook_func
        self.hyper_params = hyper_params
        self.t_obs, self.t_pred = hyper_params.obs_len, hyper_params.pred_len
        self.obs_svd, self.pred_svd = hyper_params.obs_svd, hyper_params.pred_svd
        self.k = hyper_params.k
        self.s = hyper_params.num_samples
        self.dim = hyper_params.traj_dim
        self.static_dist = hyper_params.static_dist

        self.ET_m_descriptor = ETDescriptor(hyper_params=hyper_params, norm_sca=True)
        self.ET_s_descriptor = ETDescriptor(hyper_params=hyper_params, norm_sca=False)
        self.ET_m_anchor = ETAnchor(hyper_params=hyper_params)
        self.ET_s_anchor = ETAnchor(hyper_params=hyper_params)

    # ---- scene-size fast path --------------------------------------------------------------------------------
    # The reference's real workload is one scene (N <= 57 pedestrians) per forward: kernel time is a few
    # microseconds, everything else is host overhead.  When the observations are already a contiguous fp32 tensor
    # on the device, the projection runs as ONE single-workgroup launch that also produces the mean-centred
    # obs_ori (model.py:86-89), all intermediates come from one allocation, and the C ABI is called with plain ints.
    def _scene_ok(self, obs_traj):
        return (obs_traj.is_cuda and obs_traj.dtype == torch.float32 and obs_traj.is_contiguous() and obs_traj.dim() == 3
                and 0 < obs_traj.shape[0] <= ops.L.SCENE_MAX_N and self.ET_m_descriptor.U_obs_trunc.device == obs_traj.device)

    def _scene_project(self, obs_traj):
        """-> C_obs (k,N), obs_ori (2,N), nrm (4,N): three views of one fresh (k+6,N) block."""
        n, t_obs = obs_traj.shape[0], obs_traj.shape[1]
        k = self.k
        block = torch.empty((k + 6, n), device=obs_traj.device)
        base = block.data_ptr()
        rc = ops.L.fast("et_scene_project")(
            obs_traj.data_ptr(), n, t_obs, k, self.ET_m_descriptor.U_obs_trunc.data_ptr(),
            self.ET_s_descriptor.U_obs_trunc.data_ptr(), ops.MODE_SPLIT, self.static_dist, base, base + 4 * (k + 2) * n,
            base + 4 * k * n, None, ops.L.raw_stream(obs_traj.device.index))
        if rc:
            ops.L.check(rc, "et_scene_project")
        return block[:k], block[k:k + 2], block[k + 2:]

    def _scene_project_train(self, obs_traj, pred_traj):
        """-> C_obs (k,N), obs_ori (2,N), nrm (4,N), C_gt (k,N): views of one fresh (2k+6,N) block (one launch)."""
        n, k = obs_traj.shape[0], self.k
        block = torch.empty((2 * k + 6, n), device=obs_traj.device)
        base = block.data_ptr()
        rc = ops.L.fast("et_scene_project_train")(
            obs_traj.data_ptr(), pred_traj.data_ptr(), n, obs_traj.shape[1], pred_traj.shape[1], k,
            self.ET_m_descriptor.U_obs_trunc.data_ptr(), self.ET_s_descriptor.U_obs_trunc.data_ptr(),
            self.ET_m_descriptor.U_pred_trunc.data_ptr(), self.ET_s_descriptor.U_pred_trunc.data_ptr(), ops.MODE_SPLIT,
            self.static_dist, base, base + 4 * (k + 2) * n, base + 4 * k * n, base + 4 * (k + 6) * n, None,
            ops.L.raw_stream(obs_traj.device.index))
        if rc:
            ops.L.check(rc, "et_scene_project_train")
        return block[:k], block[k:k + 2], block[k + 2:k + 6], block[k + 6:]

    def _predict(self, C_obs, obs_ori, addl_info):
        """bridge protocol of baseline/<name>/bridge.py: pre-hook -> predictor -> post-hook; returns the refinement
        coefficients (k, N, S) the predictor proposes for the scene"""
        hooks = self.hook_func
        net_in = hooks.model_forward_pre_hook(C_obs, obs_ori, addl_info)
        net_out = hooks.model_forward(net_in, self.baseline_model)
        return hooks.model_forward_post_hook(net_out, addl_info)

    def _U(self):
        return (self.ET_m_descriptor.U_obs_trunc.detach(), self.ET_m_descriptor.U_pred_trunc.detach(),
                self.ET_s_descriptor.U_obs_trunc.detach(), self.ET_s_descriptor.U_pred_trunc.detach())

    def calculate_parameters(self, obs_traj, pred_traj):
        r"""Fit both descriptors and both anchor sets on the training trajectories (model.py:34-56); run once, before
        training.  ``obs_traj`` (N, t_obs, 2) and ``pred_traj`` (N, t_pred, 2) may live on the CPU or on the GPU."""
        sd = self.static_dist
        # Descriptor initialization: one pass over ALL rows per descriptor; the kernel masks out the
        # rows of the other one (model.py:46-52 splits the tensors instead).
        grams = []
        for which in (1, 0):
            g_obs, g_pred, _ = ops.fit_gram(obs_traj, pred_traj, ops.MODE_SPLIT, sd, which)
            grams += [g_obs, g_pred]
        bases = ops.eigh_topk_batch(grams, self.k)  # the four eigenproblems side by side, one launch
        for i, desc in enumerate((self.ET_m_descriptor, self.ET_s_descriptor)):
            U_obs, U_pred = bases[2 * i][0], bases[2 * i + 1][0]
            desc.U_obs_trunc = nn.Parameter(U_obs.to(desc.U_obs_trunc.device))
            desc.U_pred_trunc = nn.Parameter(U_pred.to(desc.U_pred_trunc.device))

        # Anchor generation (model.py:55-56) on the coefficients of each descriptor's own rows
        _, U_pred_m, _, U_pred_s = self._U()
        _, C_pred, _, flag = ops.norm_project(obs_traj, pred_traj, None, U_pred_m, None, U_pred_s, ops.MODE_SPLIT, sd,
                                              want_nrm=False, want_obs=False)
        moving = flag.bool()
        C_m, C_s = C_pred[:, moving].contiguous(), C_pred[:, ~moving].contiguous()
        # the two clusterings are independent (model.py:55-56 runs them one after the other): side by side -- the static
        # one from a helper thread on a side stream (both kept for the life of the process: creating them costs more
        # than they save at these sizes), the moving one from this thread on the current stream
        dev, main = C_pred.device, torch.cuda.current_stream(C_pred.device)
        pool, side = _side_lane(dev)
        side.wait_stream(main)

        def run_static():
            with torch.cuda.device(dev), torch.cuda.stream(side):
                self.ET_s_anchor.generate_from_coefficients(C_s)

        job = pool.submit(run_static)
        try:
            self.ET_m_anchor.generate_from_coefficients(C_m)
        finally:
            job.result()  # re-raises what the helper raised
            main.wait_stream(side)

    @torch.no_grad()
    def evaluate(self, obs_traj, pred_traj, addl_info=None):
        r"""Best-of-S ADE / FDE per pedestrian without materialising ``recon_traj`` (evaluation form of
        :meth:`forward` + utils/metrics.py:73-102; utils/trainer.py:183-186 computes the same from the tensor).

        Returns:
            ade (torch.Tensor): (num_ped,), fde (torch.Tensor): (num_ped,)
        """
        sd = self.static_dist
        fast = self._scene_ok(obs_traj) and self._scene_ok(pred_traj)
        if fast:
            C_obs, obs_ori, nrm = self._scene_project(obs_traj)
        else:
            U_obs_m, U_pred_m, U_obs_s, U_pred_s = self._U()
            C_obs, _, nrm, _ = ops.norm_project(obs_traj, None, U_obs_m, None, U_obs_s, None, ops.MODE_SPLIT, sd,
                                                want_flag=False)
            obs_ori = nrm[:2] - nrm[:2].mean(dim=1, keepdim=True)
        C_pred_refine = self._predict(C_obs, obs_ori, addl_info)
        if fast and C_pred_refine.is_cuda and C_pred_refine.dtype == torch.float32 and C_pred_refine.dim() == 3:
            Cc = C_pred_refine if C_pred_refine.is_contiguous() else C_pred_refine.contiguous()
            k, n, s = Cc.shape
            out = torch.empty((2, n), device=Cc.device)
            rc = ops.L.fast("et_anchor_reconstruct_metrics")(
                Cc.data_ptr(), n, s, k, obs_traj.shape[1], pred_traj.shape[1], None, nrm.data_ptr(),
                self.ET_m_anchor.C_anchor.data_ptr(), self.ET_s_anchor.C_anchor.data_ptr(),
                self.ET_m_descriptor.U_pred_trunc.data_ptr(), self.ET_s_descriptor.U_pred_trunc.data_ptr(), ops.MODE_SPLIT,
                sd, pred_traj.data_ptr(), out.data_ptr(), out.data_ptr() + 4 * n, ops.L.raw_stream(Cc.device.index))
            if rc:
                ops.L.check(rc, "et_anchor_reconstruct_metrics")
            return out[0], out[1]
        U_obs_m, U_pred_m, U_obs_s, U_pred_s = self._U()
        A_m, A_s = self.ET_m_anchor.C_anchor.detach(), self.ET_s_anchor.C_anchor.detach()
        return ops.anchor_reconstruct_metrics(C_pred_refine.contiguous(), pred_traj, A_m, A_s, U_pred_m, U_pred_s,
                                              ops.MODE_SPLIT, sd, nrm=nrm, t_obs=obs_traj.shape[1])

    # ---- replayed scene calls ---------------------------------------------------------------------------------------
    # A scene call is four to five small launches and ~35 us of host work around them.  For loops that visit the SAME
    # device-resident scenes again and again (the reference's test loop walks the same ~70 scenes per split every epoch,
    # utils/trainer.py:170-190; with the split held on the device each scene is a view at a fixed address) the whole call
    # -- projection, the bridge's hooks, the predictor, reconstruction -- is captured ONCE per scene in a HIP graph
    # (torch.cuda.CUDAGraph; the library's launches go to the capturing stream like any other kernel) and then replayed:
    # one graph launch per call, ~21 us.  The graph reads the scene's tensors where they are (no staging copies: with them a
    # replay cost more than the eager call), so it is keyed by their addresses and keeps them alive.
    # Memory: every captured graph owns a private allocator pool (>= one 2 MB segment) plus its output buffers, so the
    # cache is bounded and evicts the least recently used scene; a test split is <= ~1000 scenes (~2-4 GB).
    _SCENE_GRAPH_LIMIT = 1024
    _PREDICTOR_STAMP_EVERY = 64

    def _predictor_stamp(self):
        """Addresses of the predictor's parameters and buffers.  Walking hundreds of tensors costs more host time than the
        ~21 us replay it protects, so the FULL walk is redone when something that can move them was seen -- this wrapper's
        ``_apply`` (.to() / .cuda() / .half()), ``load_state_dict``, ``train()`` / ``eval()``, ``baseline_model`` re-assigned
        or deleted -- and every 64th call; EVERY call compares a cheap probe: the identity of the predictor module, the
        number of its direct parameter / buffer / sub-module slots and the addresses of its first, middle and last parameter (a
        ``.to()`` / ``.half()`` / ``.cuda()`` on the sub-module behind the wrapper's back moves all of them).  A cache entry also
        keeps the tensors it captured alive (``_graph_for``: ``keep``), so a replay that slipped through reads live memory.
        Not noticed: an in-place ``.data`` swap that keeps the storage (never was)."""
        d = self.__dict__
        bm = self.baseline_model
        first = next(bm.parameters(), None)  # (stops at the first one: no walk of the module tree)
        sampled = d.get("_pstamp_sampled", ())  # the middle and the last parameter OBJECTS of the last full walk
        probe = (id(bm), len(bm._parameters), len(bm._buffers), len(bm._modules),
                 0 if first is None else first.data_ptr()) + tuple(t.data_ptr() for t in sampled)
        n = d.get("_pstamp_calls", 0)
        d["_pstamp_calls"] = n + 1
        if d.get("_pstamp") is None or d.get("_pstamp_probe") != probe or n % self._PREDICTOR_STAMP_EVERY == 0:
            ps = list(bm.parameters())
            d["_pstamp_sampled"] = sampled = tuple(ps[i] for i in sorted({len(ps) // 2, len(ps) - 1})) if ps else ()
            d["_pstamp_probe"] = probe[:5] + tuple(t.data_ptr() for t in sampled)
            d["_pstamp"] = (tuple(p.data_ptr() for p in ps), tuple(b.data_ptr() for b in bm.buffers()))
        return d["_pstamp"]

    def __setattr__(self, name, value):
        if name == "baseline_model":
            self.__dict__["_pstamp"] = None
        super().__setattr__(name, value)

    def __delattr__(self, name):
        if name == "baseline_model":
            self.__dict__["_pstamp"] = None
        super().__delattr__(name)

    def _apply(self, fn, *args, **kwargs):
        self.__dict__["_pstamp"] = None
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self.__dict__["_pstamp"] = None
        return super().load_state_dict(*args, **kwargs)

    def train(self, mode=True):
        self.__dict__["_pstamp"] = None
        return super().train(mode)

    def _graph_for(self, kind, obs_traj, pred_traj, run):
        params = (self.ET_m_descriptor.U_obs_trunc, self.ET_m_descriptor.U_pred_trunc, self.ET_s_descriptor.U_obs_trunc,
                  self.ET_s_descriptor.U_pred_trunc, self.ET_m_anchor.C_anchor, self.ET_s_anchor.C_anchor)
        key = (kind, obs_traj.data_ptr(), tuple(obs_traj.shape), 0 if pred_traj is None else pred_traj.data_ptr(),
               None if pred_traj is None else tuple(pred_traj.shape))
        # (calculate_parameters / load_state_dict may re-register the parameters, the predictor may be moved, re-allocated
        # or switched between train() and eval(): a graph holds raw pointers and the code path taken at capture time)
        stamp = (tuple(p.data_ptr() for p in params), self.training, self.baseline_model.training, self._predictor_stamp())
        cache = self.__dict__.setdefault("_scene_graphs", {})
        entry = cache.get(key)
        if entry is not None and entry["stamp"] == stamp:
            cache[key] = cache.pop(key)  # most recently used last
            return entry
        if entry is not None:
            del cache[key]  # stale capture
        while len(cache) >= self._SCENE_GRAPH_LIMIT:
            cache.pop(next(iter(cache)))  # least recently used first: its graph, pool and buffers are released
        dev = obs_traj.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # lazy initialisation (allocator pools, the predictor's own) outside the capture
            for _ in range(2):
                run(obs_traj, pred_traj)
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            outs = run(obs_traj, pred_traj)
        # `keep`: the tensors whose addresses the graph holds stay alive as long as the capture does
        keep = params + tuple(self.baseline_model.parameters()) + tuple(self.baseline_model.buffers())
        entry = cache[key] = dict(stamp=stamp, graph=graph, ins=(obs_traj, pred_traj), outs=outs, keep=keep)
        return entry

    @torch.no_grad()
    def evaluate_replayed(self, obs_traj, pred_traj):
        r""":meth:`evaluate` through a HIP graph captured per scene (keyed by the ADDRESSES and shapes of ``obs_traj`` /
        ``pred_traj``: meant for scenes that stay on the device and are visited every epoch; their contents may change).

        Needs contiguous fp32 tensors on the device, a predictor and hooks that are plain tensor code (no host
        synchronisation, no data-dependent Python control flow -- the ten bridges of the reference qualify) and no
        ``addl_info``.  The returned tensors are the graph's own output buffers: valid until the next replayed call on the
        same scene (``.clone()`` to keep them).  The cache keeps the scene's tensors alive and holds at most 1024 scenes (least
        recently used evicted: every graph owns an allocator pool of >= 2 MB); a capture is redone when the ET parameters,
        the predictor's parameters / buffers or the train / eval mode changed.  For inputs the scene path does not take the
        call is the eager one."""
        if not (self._scene_ok(obs_traj) and self._scene_ok(pred_traj)):
            return self.evaluate(obs_traj, pred_traj)
        entry = self._graph_for("evaluate", obs_traj, pred_traj, lambda o, p: self.evaluate(o, p))
        if entry is None:
            return self.evaluate(obs_traj, pred_traj)
        entry["graph"].replay()
        return entry["outs"]

    @torch.no_grad()
    def forward_replayed(self, obs_traj):
        r"""The inference form of :meth:`forward` (``pred_traj=None``) through a HIP graph captured per scene; same
        conditions and the same ownership of the returned ``recon_traj`` as :meth:`evaluate_replayed`."""
        if not self._scene_ok(obs_traj):
            return self.forward(obs_traj)
        entry = self._graph_for("forward", obs_traj, None, lambda o, p: self.forward(o))
        if entry is None:
            return self.forward(obs_traj)
        entry["graph"].replay()
        return entry["outs"]

    def forward(self, obs_traj, pred_traj=None, addl_info=None):
        r"""One scene through projection -> predictor -> anchor refinement -> reconstruction (model.py:58-125).

        ``obs_traj`` (N, t_obs, 2); ``pred_traj`` (N, t_pred, 2) only when training (adds the three loss terms);
        ``addl_info`` is handed to the bridge hooks untouched.  Returns ``{"recon_traj": (S, N, t_pred, 2)}`` plus
        ``loss_eigentraj``, ``loss_euclidean_ade``, ``loss_euclidean_fde`` when ``pred_traj`` is given.
        """
        sd = self.static_dist
        if pred_traj is None and self._scene_ok(obs_traj):  # inference on one scene: the lean path
            C_obs, obs_ori, nrm = self._scene_project(obs_traj)
            C_pred_refine = self._predict(C_obs, obs_ori, addl_info)
            if (C_pred_refine.is_cuda and C_pred_refine.dtype == torch.float32 and C_pred_refine.dim() == 3
                    and not (C_pred_refine.requires_grad and torch.is_grad_enabled())):
                Cc = C_pred_refine if C_pred_refine.is_contiguous() else C_pred_refine.contiguous()
                k, n, s = Cc.shape
                recon = torch.empty((s, n, self.t_pred, 2), device=Cc.device)
                rc = ops.L.fast("et_anchor_reconstruct_fwd")(
                    Cc.data_ptr(), n, s, k, obs_traj.shape[1], self.t_pred, None, nrm.data_ptr(),
                    self.ET_m_anchor.C_anchor.data_ptr(), self.ET_s_anchor.C_anchor.data_ptr(),
                    self.ET_m_descriptor.U_pred_trunc.data_ptr(), self.ET_s_descriptor.U_pred_trunc.data_ptr(),
                    ops.MODE_SPLIT, sd, recon.data_ptr(), ops.L.raw_stream(Cc.device.index))
                if rc:
                    ops.L.check(rc, "et_anchor_reconstruct_fwd")
                return {"recon_traj": recon}
            U_obs_m, U_pred_m, U_obs_s, U_pred_s = self._U()
            A_m, A_s = self.ET_m_anchor.C_anchor.detach(), self.ET_s_anchor.C_anchor.detach()
            recon = ops.anchor_reconstruct(C_pred_refine, A_m, A_s, U_pred_m, U_pred_s, ops.MODE_SPLIT, sd, nrm=nrm,
                                           t_obs=obs_traj.shape[1])
            return {"recon_traj": recon.to(obs_traj.device)}
        if (pred_traj is not None and self._scene_ok(obs_traj) and self._scene_ok(pred_traj)
                and pred_traj.shape[0] == obs_traj.shape[0] and self.ET_m_anchor.C_anchor.device == obs_traj.device):
            # training on one scene (utils/trainer.py:126-152): projection of obs and ground truth in one launch,
            # reconstruction + the three losses in one more, one launch for their gradient (csrc/et_train.hip)
            C_obs, obs_ori, nrm, C_gt = self._scene_project_train(obs_traj, pred_traj)
            C_pred_refine = self._predict(C_obs, obs_ori, addl_info)
            if C_pred_refine.is_cuda and C_pred_refine.dtype == torch.float32 and C_pred_refine.dim() == 3:
                Cc = C_pred_refine if C_pred_refine.is_contiguous() else C_pred_refine.contiguous()
                recon, l_e, l_ade, l_fde = _SceneLosses.apply(Cc, self, nrm, C_gt, pred_traj, obs_traj.shape[1])
                return {"recon_traj": recon, "loss_eigentraj": l_e, "loss_euclidean_ade": l_ade, "loss_euclidean_fde": l_fde}
        return self._forward_composite(obs_traj, pred_traj, addl_info)

    def _forward_composite(self, obs_traj, pred_traj=None, addl_info=None):
        """:meth:`forward` out of the general kernels and framework operators: any batch size, tensors on any device
        (they are moved), predictors that answer in another dtype.  What the scene paths above are checked against."""
        sd = self.static_dist
        U_obs_m, U_pred_m, U_obs_s, U_pred_s = self._U()
        A_m, A_s = self.ET_m_anchor.C_anchor.detach(), self.ET_s_anchor.C_anchor.detach()

        # Projection of every pedestrian with its own descriptor (model.py:73-83) + the cached
        # normaliser state; nrm[:2] is the absolute last observed position (model.py:86-88)
        C_obs, C_pred_gt, nrm, flag = ops.norm_project(
            obs_traj, pred_traj, U_obs_m, U_pred_m if pred_traj is not None else None,
            U_obs_s, U_pred_s if pred_traj is not None else None, ops.MODE_SPLIT, sd, want_flag=pred_traj is not None)
        obs_ori = nrm[:2] - nrm[:2].mean(dim=1, keepdim=True)  # move scene to origin (model.py:89)

        # the plugged-in predictor, through its bridge (model.py:93-95)
        C_pred_refine = self._predict(C_obs, obs_ori, addl_info)

        # Anchor refinement + reconstruction in one kernel (model.py:98-105)
        pred_traj_recon = ops.anchor_reconstruct(C_pred_refine, A_m, A_s, U_pred_m, U_pred_s, ops.MODE_SPLIT, sd,
                                                 nrm=nrm, t_obs=obs_traj.shape[1])
        pred_traj_recon = pred_traj_recon.to(obs_traj.device)

        output = {"recon_traj": pred_traj_recon}

        if pred_traj is not None:
            moving = flag.bool()[None, :, None]
            C_pred = torch.where(moving, A_m[:, None, :], A_s[:, None, :]) + C_pred_refine.to(A_m.device)
            C_pred_gt = C_pred_gt.detach()  # low-rank approximation of the gt trajectory (model.py:113-116)
            gt = pred_traj.to(pred_traj_recon.device)

            # model.py:119-123: best-of-S distances in coefficient space and in metres (mean / final step)
            coef_err = torch.linalg.vector_norm(C_pred - C_pred_gt[:, :, None], dim=0)        # (N, S)
            step_err = torch.linalg.vector_norm(pred_traj_recon - gt[None], dim=-1)           # (S, N, t_pred)
            output["loss_eigentraj"] = coef_err.amin(dim=-1).mean()
            output["loss_euclidean_ade"] = step_err.mean(dim=-1).amin(dim=0).mean()
            output["loss_euclidean_fde"] = step_err[:, :, -1].amin(dim=0).mean()

        return output
