import sys, torch
sys.path.insert(0, '.')
from iseg_amd import functional as F, nn
from oracle import tf_ops as O
nn.set_device("cuda:0")
for C, shape in ((16,(2,4,4)),(64,(2,16,16)),(64,(2,4,4))):
    x=torch.randn(*shape,C)
    g=torch.nn.Parameter((torch.rand(C)+0.5).cuda()); b=torch.nn.Parameter((torch.randn(C)*0.1).cuda())
    mm=(torch.randn(C)*0.1).cuda(); mv=(torch.rand(C)+0.5).cuda()
    for relu in (False, True):
        xg=x.cuda().requires_grad_(True)
        x1=F.replace_nan_or_inf(xg)
        y=F.batch_norm(x1,g,b,mm,mv,1e-3,0.9,False,relu=relu)
        z=F.add(y, y)
        dy=torch.randn(*shape,C)
        z.backward(dy.cuda())
        xr=x.double().requires_grad_(True)
        yr=O.batch_norm_infer(xr,g.detach().cpu().double(),b.detach().cpu().double(),mm.cpu().double(),mv.cpu().double(),1e-3)
        if relu: yr=torch.relu(yr)
        (yr+yr).backward(dy.double())
        print(C, shape, relu, (y.detach().cpu().double()-yr).abs().max().item(), (xg.grad.cpu().double()-xr.grad).abs().max().item(), xr.grad.abs().max().item())
