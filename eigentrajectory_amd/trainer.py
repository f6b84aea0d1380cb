"""Training / evaluation harness around the wrapper.

A compact counterpart of the reference's utils/trainer.py (ETTrainer and its two batching modes),
written against :class:`eigentrajectory_amd.EigenTrajectory`:

* ``init_descriptor``   utils/trainer.py:48-55: train+val trajectories, y-flip augmentation, fit of the
                        descriptors and anchors (``model.calculate_parameters``);
* optimiser / schedule  AdamW(lr, weight_decay) + StepLR(lr_schd_step, lr_schd_gamma) over the
                        predictor's parameters, gradient-norm clipping (utils/trainer.py:39-46, 141-144);
* ``train`` / ``valid`` loss = loss_eigentraj + loss_euclidean_ade + loss_euclidean_fde with NaN -> 0
                        (utils/trainer.py:132-134); "collated" batches (scenes concatenated until
                        ``batch_size`` pedestrians, one forward: :211-231) or "sequenced" ones (one scene per
                        forward, gradients accumulated over the batch: :120-154) for the graph predictors;
* ``test``              best-of-S ADE / FDE over the test scenes (:173-195) through the fused epilogue.

Data parallelism: with ``torch.distributed`` initialised (one process per GPU, backend "nccl" = RCCL),
the predictor is wrapped in DistributedDataParallel and every rank takes every ``world``-th batch.  The
descriptor parameters receive no gradient (they are detached in the forward, like in the reference), so
only the predictor's gradients cross xGMI.
"""
from __future__ import annotations

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
    def _batches(self, data, shuffle, drop_last, seed=0):
        gen = torch.Generator().manual_seed(seed) if shuffle else None
        batches = list(scene_batches(data.num_peds_in_seq, self.hp.batch_size, shuffle, drop_last, gen))
        return batches[self.rank::self.world]

    def _addl(self, scene_mask):
        return {"scene_mask": scene_mask.to(self.device), "num_samples": self.hp.num_samples}

    def _loss(self, obs, pred, addl):
        out = self.model(obs, pred, addl_info=addl)
        loss = out["loss_eigentraj"] + out["loss_euclidean_ade"] + out["loss_euclidean_fde"]
        return torch.where(torch.isnan(loss), torch.zeros_like(loss), loss)  # utils/trainer.py:133

    def _run_batch(self, data, batch, train):
        if self.mode == "collated":
            obs, pred, mask, _ = collate_scenes(data, batch)
            loss = self._loss(obs.to(self.device), pred.to(self.device), self._addl(mask))
            if train:
                loss.backward()
            return float(loss.item())
        total = 0.0
        for idx in batch:  # sequenced: one scene per forward, gradients accumulate over the batch
            obs, pred = data[idx]
            n = obs.shape[0]
            loss = self._loss(obs.to(self.device), pred.to(self.device),
                              self._addl(torch.ones((n, n), dtype=torch.bool))) / len(batch)
            if train:
                loss.backward()
            total += float(loss.item())
        return total

    # -------------------------------------------------------------------------------- train / eval
    def train(self, epoch=0):
        self.model.train()
        losses = []
        for batch in self._batches(self.train_data, shuffle=True, drop_last=True, seed=epoch):
            self.optimizer.zero_grad(set_to_none=True)
            losses.append(self._run_batch(self.train_data, batch, train=True))
            if self.hp.clip_grad is not None:
                torch.nn.utils.clip_grad_norm_(self.predictor.parameters(), self.hp.clip_grad)
            self.optimizer.step()
        if self.scheduler is not None:
            self.scheduler.step()
        self.log["train_loss"].append(float(sum(losses) / max(len(losses), 1)))
        return self.log["train_loss"][-1]

    @torch.no_grad()
    def valid(self):
        self.model.eval()
        losses = [self._run_batch(self.val_data, b, train=False) for b in self._batches(self.val_data, False, False)]
        t = torch.tensor([sum(losses), float(len(losses))], device=self.device, dtype=torch.float64)
        if self.world > 1:
            dist.all_reduce(t)
        self.log["val_loss"].append(float(t[0] / max(float(t[1]), 1.0)))
        return self.log["val_loss"][-1]

    def fit(self, epochs):
        best = math.inf
        best_state = None
        for epoch in range(epochs):
            self.train(epoch)
            val = self.valid()
            if val < best:  # utils/trainer.py:75-79 keeps the best-validation weights
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
        ades, fdes = [], []
        for idx in range(len(data))[self.rank::self.world]:
            obs, pred = data[idx]
            n = obs.shape[0]
            a, f = self.model.evaluate(obs.to(self.device), pred.to(self.device),
                                       self._addl(torch.ones((n, n), dtype=torch.bool)))
            ades.append(a)
            fdes.append(f)
        t = torch.stack([torch.cat(ades).double().sum(), torch.cat(fdes).double().sum(),
                         torch.tensor(float(sum(x.numel() for x in ades)), device=self.device, dtype=torch.float64)])
        if self.world > 1:
            dist.all_reduce(t)
        return {"ADE": float(t[0] / t[2]), "FDE": float(t[1] / t[2])}
