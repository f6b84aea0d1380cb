import sys, numpy as np, torch
sys.path.insert(0,'.')
from tests import _golden as G
from eigentrajectory_amd import EigenTrajectory, ops
from eigentrajectory_amd.utils import default_hyper_params, DotDict
from oracle import et_oracle as eo
dev=torch.device('cuda:0')
obs,pred=G.eth_fit_input()
hp=default_hyper_params(static_dist=G.static_dist('eth'))
m=EigenTrajectory(torch.nn.Identity(), DotDict(), hp).to(dev)
m.calculate_parameters(torch.from_numpy(obs).to(dev), torch.from_numpy(pred).to(dev))
print('inertia m', m.ET_m_anchor.inertia_, 's', m.ET_s_anchor.inertia_)
_,Upm,_,Ups=m._U()
_,C,_,flag=ops.norm_project(torch.from_numpy(obs).to(dev), torch.from_numpy(pred).to(dev), None,Upm,None,Ups,2,hp.static_dist,want_nrm=False,want_obs=False)
x=C[:,flag.bool()].contiguous()
print(x.shape, x.abs().max().item())
np.random.seed(0); first=np.random.randint(x.shape[1]); print('first',first)
c0=ops.kmeans_init_farthest(x,20,first)
res=ops.kmeans_fit(x,c0,100,1e-4)
print(res['n_iter'],res['inertia'],res['error'], res['trace'][:3])
r0,_=eo.kmeans_init_farthest(x.cpu().numpy(),20,first)
ref=eo.kmeans_fit(x.cpu().numpy(),r0,100,1e-4)
print(ref['n_iter'],ref['inertia'], np.array_equal(ref['labels'],res['labels'].cpu().numpy()))
A=m.ET_m_anchor.C_anchor.detach()
print('A vs res centroids equal', torch.equal(A,res['centroids']))
lb,ms=ops.kmeans_predict(x,A); print('inertia via predict', (-ms).mean().item())
