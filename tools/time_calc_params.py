import sys, time, os, torch
sys.path.insert(0, '/root/repo')
from eigentrajectory_amd import EigenTrajectory
from eigentrajectory_amd.synth import synthetic_trajectories_torch
from eigentrajectory_amd.utils import DotDict, default_hyper_params
dev = torch.device("cuda:0")
class Zero(torch.nn.Module):
    def forward(self, x): return torch.zeros((6, x.size(1), 20), device=x.device)
hooks = DotDict(model_forward_pre_hook=lambda c, o, a=None: torch.cat([c, o], dim=0), model_forward=lambda x, m: m(x), model_forward_post_hook=lambda y, a=None: y)
model = EigenTrajectory(Zero(), hooks, default_hyper_params(static_dist=0.3)).to(dev)
o, p = synthetic_trajectories_torch(70_316, dev, seed=6)
for _ in range(3): model.calculate_parameters(o, p)
torch.cuda.synchronize()
ts = []
for _ in range(7):
    t0 = time.perf_counter(); model.calculate_parameters(o, p); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(os.environ.get("ET_OPT_KMEANS_LOOP", "auto"), "calculate_parameters ms:", ["%.2f" % (t * 1e3) for t in sorted(ts)])
