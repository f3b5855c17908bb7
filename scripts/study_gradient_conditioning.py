#!/usr/bin/env python3
"""CPU study (oracle): how well-conditioned is the nerfacto training gradient at the fp32 level? At the benchmark's
configuration (full tables ~ N(0, 0.3), 1024 rays) the exact float64 gradient is evaluated at the parameters and at the
parameters moved by half an fp32 ulp (random sign per element), next to the reference's own fp32 evaluation. Output:
profiles/r03_gradient_conditioning.txt. The per-tensor tolerance of tests/test_gpu_bench_parity.py rests on it."""
import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from oracle import nerfacto_oracle as orc
import bench
torch.set_num_threads(8)
cfg=orc.NerfactoCfg()
n=1024
o,d,cam,tgt=(torch.from_numpy(a[:n]) for a in bench.synthetic_rays(1003))
rs=np.random.RandomState(0)
jit=[torch.from_numpy(rs.uniform(0,1,(n,1))) for _ in range(3)]
base=orc.init_params(cfg,seed=0,table_std=0.3)
def run(params, dtype):
    p={k:v.detach().to(dtype).clone().requires_grad_(True) for k,v in params.items()}
    out=orc.nerfacto_forward(p,cfg,o.to(dtype),d.to(dtype),cam[:,0],[j.to(dtype) for j in jit],training=True,proposal_requires_grad=False)
    ld=orc.nerfacto_losses(out,tgt.to(dtype),cfg); sum(ld.values()).backward()
    return {k:(v.grad.double().numpy() if v.grad is not None else None) for k,v in p.items()}
t=time.time()
g64=run(base,torch.float64); print('f64',time.time()-t)
g32=run(base,torch.float32)
gen=torch.Generator().manual_seed(1)
pert={k:(v.double()*(1+ (torch.randint(0,2,v.shape,generator=gen).double()*2-1)*2.0**-24)) for k,v in base.items()}
g64p=run(pert,torch.float64)
rel=lambda a,b: np.linalg.norm(a-b)/max(np.linalg.norm(b),1e-30)
for k in g64:
    if g64[k] is None or not k.startswith('field'): continue
    if 'hash_table' in k:
        T=1<<19
        for l in (0,5,10,15):
            s=slice(l*T,(l+1)*T); print(k,l,'ref32-f64 %.2e  perturbed-f64 %.2e'%(rel(g32[k][s],g64[k][s]),rel(g64p[k][s],g64[k][s])))
    else:
        print(k,'ref32-f64 %.2e  perturbed-f64 %.2e'%(rel(g32[k],g64[k]),rel(g64p[k],g64[k])))
