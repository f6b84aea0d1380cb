"""Training / evaluation harness around the wrapper.

A compact counterpart of the reference's utils/trainer.py (ETTrainer and its two batching strategies),
written against :class:`eigentrajectory_amd.EigenTrajectory`:

* ``init_descriptor``   utils/trainer.py:48-55: train+val trajectories, y-flip augmentation, fit of the
                        descriptors and anchors (``model.calculate_parameters``);
* optimiser / schedule  AdamW(lr, weight_decay) + StepLR(lr_schd_step, lr_schd_gamma), stepped once per epoch
                        after validation (utils/trainer.py:39-46, 68-72), gradient-norm clipping (:141-144);
* ``train``             loss = loss_eigentraj + loss_euclidean_ade + loss_euclidean_fde with NaN -> 0 (:132-134).
                        "collated" (:211-231): scenes concatenated until a batch holds ``batch_size`` pedestrians
                        (shuffled, last incomplete batch dropped), one forward / backward / step per batch.
                        "sequenced" (:120-154, the graph predictors): scenes in dataset order, one forward per
                        scene, the losses of ``batch_size`` consecutive SCENES summed and divided by
                        ``batch_size`` (also for the shorter last group), one step per group;
* ``valid``             pedestrian-weighted ``loss_euclidean_fde`` (:156-170, :233-247) -- the number that selects
                        the best checkpoint (:74-75);
* ``test``              best-of-S ADE / FDE over the test scenes (:173-195) through the fused epilogue.

Data parallelism: with ``torch.distributed`` initialised (one process per GPU, backend "nccl" = RCCL), the
predictor is wrapped in DistributedDataParallel and rank r takes batches r, r + world, ... of every epoch.
Every rank runs the SAME number of optimiser steps (the tail that does not fill a round of ``world`` batches is
dropped, like the reference drops its incomplete last batch), and in sequenced mode only the last backward of a
group synchronises gradients (``no_sync`` on the others), so the ranks' collectives always pair up.  Gradients
are averaged over ranks: a sequenced run on ``world`` ranks equals a single-process run with
``batch_size * world``.  Validation and test run on the bare predictor (no collectives in the forward) and
reduce their sums once at the end.  The descriptor parameters receive no gradient (they are detached in the
forward, like in the reference), so only the predictor's gradients cross xGMI.
"""
from __future__ import annotations

import contextlib
import math

import torch
import torch.distributed as dist

from .data import collate_scenes, scene_batches
from .utils import augment_trajectory


class ETTrainer:
    def __init__(self, model, hyper_params, train_data, val_data, test_data=None, mode="collated", device=None):
        assert mode in ("collated", "sequenced")
        self.hp = hyper_params
        self.mode = mode
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.model = model.to(self.device)
        self.train_data, self.val_data, self.test_data = train_data, val_data, test_data
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.predictor = self.model.baseline_model
        if self.world > 1 and any(p.requires_grad for p in self.predictor.parameters()):
            self.model.baseline_model = torch.nn.parallel.DistributedDataParallel(self.predictor, device_ids=[self.device.index])
        params = [p for p in self.predictor.parameters() if p.requires_grad]
        self.optimizer = torch.optim.AdamW(params, lr=hyper_params.lr, weight_decay=hyper_params.weight_decay) if params else None
        self.scheduler = (torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=hyper_params.lr_schd_step,
                                                          gamma=hyper_params.lr_schd_gamma)
                          if self.optimizer is not None and hyper_params.lr_schd else None)
        self.log = {"train_loss": [], "val_loss": []}

    # ------------------------------------------------------------------------------ descriptor fit
    def init_descriptor(self):
        obs = torch.cat([self.train_data.obs_traj, self.val_data.obs_traj], dim=0)
        pred = torch.cat([self.train_data.pred_traj, self.val_data.pred_traj], dim=0)
        obs, pred = augment_trajectory(obs, pred)
        self.model.calculate_parameters(obs.to(self.device), pred.to(self.device))

    # ------------------------------------------------------------------------------------ batching
    def _shard(self, batches, equal_steps):
        """Batches of this rank: r, r + world, ...; with ``equal_steps`` every rank gets the same number."""
        if equal_steps:
            batches = batches[:len(batches) // self.world * self.world]
        return batches[self.rank::self.world]

    def _batches(self, data, train, seed=0):
        if self.mode == "collated":
            gen = torch.Generator().manual_seed(seed) if train else None  # same permutation on every rank
            batches = list(scene_batches(data.num_peds_in_seq, self.hp.batch_size, shuffle=train, drop_last=train,
                                         generator=gen))
        else:  # sequenced: batch_size consecutive scenes, dataset order (the reference's DataLoader(batch_size=1))
            b = int(self.hp.batch_size)
            batches = [list(range(i, min(i + b, len(data)))) for i in range(0, len(data), b)]
        return self._shard(batches, equal_steps=train)

    def _addl(self, scene_mask):
        return {"scene_mask": scene_mask.to(self.device), "num_samples": self.hp.num_samples}

    def _forward(self, obs, pred, addl):
        return self.model(obs.to(self.device), pred.to(self.device), addl_info=addl)

    @staticmethod
    def _train_loss(out):
        loss = out["loss_eigentraj"] + out["loss_euclidean_ade"] + out["loss_euclidean_fde"]
        return torch.where(torch.isnan(loss), torch.zeros_like(loss), loss)  # utils/trainer.py:133

    def _scene(self, data, idx):
        obs, pred = data[idx]
        n = obs.shape[0]
        return obs, pred, self._addl(torch.ones((n, n), dtype=torch.bool))

    def _train_batch(self, data, batch):
        """One optimiser step's worth of forward/backward.  -> the step's loss (float)."""
        if self.mode == "collated":
            obs, pred, mask, _ = collate_scenes(data, batch)
            loss = self._train_loss(self._forward(obs, pred, self._addl(mask)))
            loss.backward()
            return float(loss.item())
        total = 0.0
        ddp = self.model.baseline_model if self.model.baseline_model is not self.predictor else None
        for j, idx in enumerate(batch):  # gradients accumulate over the group; only the last scene all-reduces them
            # DDP decides in the FORWARD whether the coming backward synchronises, so the whole scene is inside no_sync
            with (ddp.no_sync() if ddp is not None and j + 1 < len(batch) else contextlib.nullcontext()):
                loss = self._train_loss(self._forward(*self._scene(data, idx))) / float(self.hp.batch_size)
                loss.backward()
            total += float(loss.item())
        return total

    def _reduce(self, *values):
        t = torch.tensor(values, device=self.device, dtype=torch.float64)
        if self.world > 1:
            dist.all_reduce(t)
        return t.tolist()

    class _bare_predictor:
        """Evaluation runs on the un-wrapped predictor: DDP's forward may broadcast buffers, and the ranks do not
        run the same number of evaluation forwards."""

        def __init__(self, trainer):
            self.t = trainer

        def __enter__(self):
            self.saved = self.t.model.baseline_model
            self.t.model.baseline_model = self.t.predictor

        def __exit__(self, *exc):
            self.t.model.baseline_model = self.saved

    # -------------------------------------------------------------------------------- train / eval
    def train(self, epoch=0):
        if self.optimizer is None:
            raise RuntimeError("the predictor has no trainable parameters: nothing to train")
        self.model.train()
        loss_sum, n_units = 0.0, 0
        for batch in self._batches(self.train_data, train=True, seed=epoch):
            self.optimizer.zero_grad(set_to_none=True)
            loss_sum += self._train_batch(self.train_data, batch)
            n_units += 1 if self.mode == "collated" else len(batch)  # the reference divides by len(loader)
            if self.hp.clip_grad is not None:
                torch.nn.utils.clip_grad_norm_(self.predictor.parameters(), self.hp.clip_grad)
            self.optimizer.step()
        loss_sum, n_units = self._reduce(loss_sum, n_units)
        if self.mode == "sequenced":
            loss_sum /= self.world  # a group's loss is divided by batch_size per rank; the effective group is world x larger
        self.log["train_loss"].append(loss_sum / max(n_units, 1.0))
        return self.log["train_loss"][-1]

    @torch.no_grad()
    def valid(self):
        """val_loss = sum over batches of loss_euclidean_fde * num_ped / total num_ped (utils/trainer.py:156-170)."""
        self.model.eval()
        fde_sum, n_ped = 0.0, 0
        with self._bare_predictor(self):
            for batch in self._batches(self.val_data, train=False):
                if self.mode == "collated":
                    obs, pred, mask, _ = collate_scenes(self.val_data, batch)
                    groups = [(obs, pred, self._addl(mask))]
                else:
                    groups = [self._scene(self.val_data, idx) for idx in batch]
                for obs, pred, addl in groups:
                    fde_sum += float(self._forward(obs, pred, addl)["loss_euclidean_fde"].item()) * obs.shape[0]
                    n_ped += obs.shape[0]
        fde_sum, n_ped = self._reduce(fde_sum, n_ped)
        self.log["val_loss"].append(fde_sum / max(n_ped, 1.0))
        return self.log["val_loss"][-1]

    def fit(self, epochs):
        best = math.inf
        best_state = None
        for epoch in range(epochs):
            self.train(epoch)
            val = self.valid()
            if self.scheduler is not None:
                self.scheduler.step()  # utils/trainer.py:68-69
            if epoch == 0 or val < best:  # utils/trainer.py:71-72 keeps the best-validation weights
                best = val
                best_state = {k: v.detach().clone() for k, v in self.state_dict().items()}
        return best_state

    def state_dict(self):
        """Reference-compatible keys (the DDP wrapper's ``module.`` prefix is stripped)."""
        return {k.replace("baseline_model.module.", "baseline_model."): v for k, v in self.model.state_dict().items()}

    @torch.no_grad()
    def test(self, data=None):
        """-> dict(ADE, FDE) over all pedestrians of the test scenes (scene by scene, like utils/trainer.py:173-195)."""
        data = data or self.test_data
        self.model.eval()
        ade_sum = torch.zeros((), device=self.device, dtype=torch.float64)
        fde_sum = torch.zeros((), device=self.device, dtype=torch.float64)
        n_ped = 0
        with self._bare_predictor(self):
            for idx in range(len(data))[self.rank::self.world]:
                obs, pred, addl = self._scene(data, idx)
                a, f = self.model.evaluate(obs.to(self.device), pred.to(self.device), addl)
                ade_sum += a.double().sum()
                fde_sum += f.double().sum()
                n_ped += a.numel()
        a, f, n = self._reduce(float(ade_sum), float(fde_sum), n_ped)
        return {"ADE": a / max(n, 1.0), "FDE": f / max(n, 1.0)}
