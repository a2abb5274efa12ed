import sys; sys.path.insert(0,'/root/repo')
import torch
from arcnerf_amd.ops.autograd import SdfMlpJacFn
torch.manual_seed(0)
S,K,H,O=5000,32,64,20
beta=100.0
f0=(torch.randn(S,K)*0.3).cuda(); w10=(torch.randn(H,K)/K**0.5*0.3).cuda(); w20=(torch.randn(O,H)/H**0.5).cuda()
go=torch.randn(S,O).cuda(); gj=torch.randn(S,K).cuda()
res={}
for name in ('hip','torch'):
    f=f0.clone().requires_grad_(True); w1=w10.clone().requires_grad_(True); w2=w20.clone().requires_grad_(True)
    if name=='hip':
        out,jac=SdfMlpJacFn.apply(f,w1,w2,beta)
    else:
        out=torch.nn.functional.softplus(f@w1.t(),beta=beta)@w2.t()
        jac,=torch.autograd.grad(out[:,0].sum(),f,create_graph=True)
    loss=(out*go).sum()+(jac*gj).sum()
    g=torch.autograd.grad(loss,(f,w1,w2))
    res[name]=(out.detach(),jac.detach())+g
for a,b,n in zip(res['hip'],res['torch'],('out','jac','df','dw1','dw2')):
    print(n, ((a-b).abs().max()/b.abs().max()).item())
