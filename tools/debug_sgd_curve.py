"""per-variable weight-movement discrepancy of the SGD curve test (diagnostic)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import models as OM
from oracle import tf_ops as O
from tests.test_model_gpu import _setup
from iseg_amd.core_optimizer import get_optimizer
from iseg_amd.data import synthetic_batch
from iseg_amd.distribution.distribution_utils import Strategy
from iseg_amd.trainer import TrainableModel

model = _setup(torch.float32)
x, y = synthetic_batch(2, 64, 64, seed=9)
lr0 = float(sys.argv[2]) if len(sys.argv) > 2 else 2e-2
opt = get_optimizer(Strategy(one_device=True), initial_lr=lr0, end_lr=0.0, epoch_steps=10, train_epoch=1, optimizer="sgd", sgd_momentum_rate=0.9)
tm = TrainableModel(model, optimizer=opt, loss=model.custom_losses(21, 255, 2), loss_weights=model.custom_losses_weights(), metrics=model.custom_metrics(21, 255))
params = {p.iseg_name: p for p in model.parameters()}
w0 = {k: v.detach().cpu().double().clone() for k, v in params.items()}
oracle = OM.ConvNeXtASPPSGDSteps(OM.export_weights(model), x.double(), y, list(params),
                                 lambda s: O.warmup_poly_decay(s, lr0, 10, end_lr=0.0, warmup_steps=0, warmup_lr=0.0, power=0.9), momentum=0.9)
xc, yc = x.cuda(), y.cuda()
for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    lo = oracle.forward_backward()[0]
    lg = float(tm.train_step(xc, yc)[0])
    rows = []
    for k, p in params.items():
        gr = oracle._pending[0][k]
        g = p.grad.detach().cpu().double().reshape(gr.shape)
        rows.append(((g - gr).norm().item(), gr.norm().item(), k))
    rows.sort(reverse=True)
    print("step", step, "loss", lo, lg, "lr", oracle.lr_fn(step), opt.current_lr())
    for r in rows[:2]:
        print("   grad diff %.3e of %.3e  %s" % r)
    oracle.apply()
    rows = []
    for k, p in params.items():
        d = (p.detach().cpu().double().reshape(oracle.w[k].shape) - oracle.w[k]).norm().item()
        mv = (oracle.w[k] - w0[k].reshape(oracle.w[k].shape)).norm().item()
        rows.append((d, mv, k))
    rows.sort(reverse=True)
    for r in rows[:2]:
        print("   weight diff %.3e of movement %.3e  %s" % r)
    num = sum(r[0] ** 2 for r in rows) ** 0.5
    den = sum(r[1] ** 2 for r in rows) ** 0.5
    print("   TOTAL weight diff / movement", num / den, "loss rel", abs(lo - lg) / lo)
