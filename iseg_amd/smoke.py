"""One tiny train step of the flagship composition (ConvNeXt-T + ASPP) on cuda:0, checked against the CPU oracle."""
import torch


def run_smoke():
    from . import nn
    from .core_env import common_env_setup
    from .core_optimizer import get_optimizer
    from .core_train import CoreTrain
    from .data import synthetic_batch
    from .heads import convnext_tiny_aspp
    from .modelhelper import model_common_setup

    assert torch.cuda.is_available(), "smoke() needs a GPU"
    strategy = common_env_setup(use_one_device_strategy=True, mixed_precision=False, random_seed=0)
    model = convnext_tiny_aspp(num_class=21, build_input_size=(64, 64), drop_path_rate=0.0, dropout_rate=0.0, layer_scale_init_value=1.0)
    helper = model_common_setup(model, restore_checkpoint=False)
    helper.set_optimizer(get_optimizer(strategy, initial_lr=1e-3, optimizer="adamw", epoch_steps=10, train_epoch=1))
    trainer = CoreTrain(helper, None).create_trainable_model(21, batch_size=2)
    x, y = synthetic_batch(2, 64, 64, seed=0)
    x, y = x.cuda(), y.cuda()
    # ---- forward parity against the oracle (fp32 storage) ----
    from oracle import models as OM

    weights = OM.export_weights(model)
    with torch.no_grad():
        logits = model(x, training=False)[0]
    ref = OM.convnext_aspp_forward(weights, x.cpu().double(), training=False)["logits"]
    err = (logits.cpu().double() - ref).abs().max().item()
    same = torch.equal(logits.argmax(-1).cpu(), ref.argmax(-1))
    print(f"smoke: forward max|logit err| = {err:.3e}, argmax identical = {same}")
    assert err < 1e-3, err
    l0 = float(trainer.train_step(x, y)[0])
    for _ in range(3):
        l1 = float(trainer.train_step(x, y)[0])
    torch.cuda.synchronize()
    print(f"smoke: loss {l0:.4f} -> {l1:.4f}")
    assert l1 == l1 and l1 < l0 + 1e-3, (l0, l1)
    print("smoke OK")
