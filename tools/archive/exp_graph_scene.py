"""Experiment: one scene call (evaluate / forward) captured in a HIP graph through torch.cuda.CUDAGraph and replayed --
what is left of the 31-37 us per call (host + launch overhead of 4 small kernels) when the host only replays?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigentrajectory_amd import EigenTrajectory  # noqa: E402
from eigentrajectory_amd.synth import synthetic_trajectories_torch  # noqa: E402
from eigentrajectory_amd.utils import DotDict, default_hyper_params  # noqa: E402

dev = torch.device("cuda:0")


class Zero(torch.nn.Module):
    def forward(self, x):
        return torch.zeros((6, x.size(1), 20), device=x.device)


hooks = DotDict(model_forward_pre_hook=lambda c, o, a=None: torch.cat([c, o], dim=0), model_forward=lambda x, m: m(x),
                model_forward_post_hook=lambda y, a=None: y)
model = EigenTrajectory(Zero(), hooks, default_hyper_params(static_dist=0.3)).to(dev)
with torch.no_grad():
    for d_ in (model.ET_m_descriptor, model.ET_s_descriptor):
        d_.U_obs_trunc.normal_()
        d_.U_pred_trunc.normal_()
    model.ET_m_anchor.C_anchor.normal_()
    model.ET_s_anchor.C_anchor.normal_()
    obs, pred = synthetic_trajectories_torch(57, dev, seed=5)
    reps = 2000

    def timeit(fn):
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    eager = timeit(lambda: model.evaluate(obs, pred))
    ade0, fde0 = model.evaluate(obs, pred)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            model.evaluate(obs, pred)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ade, fde = model.evaluate(obs, pred)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(ade, ade0) and torch.equal(fde, fde0)
    graphed = timeit(g.replay)
    obs2 = obs.clone()

    def with_copy():
        obs.copy_(obs2)
        pred.copy_(pred)
        g.replay()
    graphed_copy = timeit(with_copy)
    print(f"evaluate per scene: eager {eager:.1f} us, graph replay {graphed:.1f} us, replay + two input copies {graphed_copy:.1f} us")
