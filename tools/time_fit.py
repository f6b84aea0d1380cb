"""Wall time of EigenTrajectory.calculate_parameters (descriptor fit + anchor clustering) on the ETH fit set, run on
the GPU box: python tools/time_fit.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import _golden as G
from eigentrajectory_amd import EigenTrajectory
from eigentrajectory_amd.utils import DotDict, default_hyper_params

dev = torch.device("cuda:0")
hooks = DotDict(model_forward_pre_hook=lambda c, o, a=None: torch.cat([c, o], 0), model_forward=lambda i, b: b(i),
                model_forward_post_hook=lambda o, a=None: o)
for scene in ("eth", "univ"):
    obs, pred = (torch.from_numpy(a).to(dev) for a in G.fit_input(scene))
    for mode in (None, "farthest"):
        hp = default_hyper_params(static_dist=G.static_dist(scene), **({"anchor_init": mode} if mode else {}))
        model = EigenTrajectory(torch.nn.Identity(), hooks, hp).to(dev)
        model.calculate_parameters(obs, pred)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            model.calculate_parameters(obs, pred)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print(f"{scene} N={obs.shape[0]} anchors={mode or 'sklearn recipe'}: calculate_parameters {min(ts) * 1e3:.1f} ms")
