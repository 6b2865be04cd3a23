import sys, torch
sys.path.insert(0, "/root/repo")
from oracle import host_threads; host_threads.apply()
import tests.test_configs_gpu as T
from oracle import models as OM
from iseg_amd import heads, nn, functional as F
from iseg_amd.data import synthetic_batch
nn.set_compute_dtype(torch.float32); nn.set_device("cuda:0")
size = (512, 512)
model = T._prep(heads.intern_image_base_aspp(build_input_size=size, dropout_rate=0.0), seed=8)
x, y = synthetic_batch(1, size[0], size[1], seed=15)
names = ["patch_embed/conv1/kernel", "block/0/layer/1/dcn/offset/kernel", "block/0/layer/1/dcn/mask/kernel", "block/0/layer/1/dcn/input_proj/kernel", "block/2/layer/10/mlp/fc1/kernel", "block/2/layer/20/gamma1",
         "block/1/downsample/conv/kernel", "block/3/layer/2/dcn/offset/kernel", "aspp_head/aspp/asp_convs_6/conv/kernel", "seg/logits_conv/kernel"]
w = OM.export_weights(model)
names = [n for n in names if n in w]
params = {p.iseg_name: p for p in model.parameters()}
for p in params.values(): p.grad = None
wg = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in w.items()}
ref = OM.intern_image_aspp_forward(wg, x.double(), training=False)["logits"]
OM.mean_ce_loss(ref, y).backward()
logits = model(x.cuda(), training=False)[0]
F.softmax_ce_mean(logits, y.cuda(), 21, 255).backward()
for n in names:
    g = params[n].grad.detach().cpu().double(); gr = wg[n].grad
    print(f"{n:45s} max-err/max {((g-gr).abs().max()/gr.abs().max()).item():.2e}  L2 rel {((g-gr).norm()/gr.norm()).item():.2e}  |g|max {gr.abs().max().item():.2e}")
# the same gradients through the ORACLE in float32: how much of the distance to fp64 is fp32 arithmetic as such
w32 = {k: v.float() for k, v in w.items()}
wg32 = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in w32.items()}
ref32 = OM.intern_image_aspp_forward(wg32, x.float(), training=False)["logits"]
OM.mean_ce_loss(ref32, y).backward()
for n in names:
    g = wg32[n].grad.double(); gr = wg[n].grad
    print(f"oracle fp32 vs fp64  {n:45s} max-err/max {((g-gr).abs().max()/gr.abs().max()).item():.2e}  L2 rel {((g-gr).norm()/gr.norm()).item():.2e}")
