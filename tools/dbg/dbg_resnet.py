import sys, torch
sys.path.insert(0, '.')
from tests.test_resnet_gpu import _build
from oracle import models as OM
blocks=(2,2,2,2)
shape=(4,96,96,3)
bb=_build(torch.float32,8,shape,blocks)
g=torch.Generator().manual_seed(1)
x=torch.randn(shape,generator=g)
TR=(len(sys.argv)<3 or sys.argv[2]=="1")
eps=bb(x.cuda(),training=TR)
w={k:(v.requires_grad_(True) if not k.endswith(("moving_mean","moving_variance")) else v) for k,v in OM.export_weights(bb).items()}
ref=OM.resnet_forward(w,x.double(),num_of_blocks=blocks,output_stride=8,training=TR)
dys=[torch.randn(tuple(e.shape),generator=g) for e in ref]
which=int(sys.argv[1]) if len(sys.argv)>1 else -1
if which>=0:
    dys=[d if i==which else torch.zeros_like(d) for i,d in enumerate(dys)]
torch.autograd.backward(list(eps),[d.cuda() for d in dys])
torch.autograd.backward(ref,[d.double() for d in dys])
gmax=max(v.grad.abs().max().item() for k,v in w.items() if v.requires_grad and v.grad is not None)
for p in bb.parameters():
    r=w[p.iseg_name].grad
    e=(p.grad.cpu().double()-r).abs().max().item()/max(r.abs().max().item(),1e-3*gmax)
    print(f"{e:9.2e} {p.iseg_name}")
