"""A tiny invocation of the flagship hot path (ConvNeXt-T + ASPP through CoreTrain's compiled model) on cuda:0, checked against
the CPU oracle: forward logits / argmax masks, then the loss curve of four AdamW steps.

Why a curve and not "the loss went down": with lr 1e-3 on two 64x64 images the restatement itself climbs for the first steps
(3.098 -> 3.25 -> 3.32 -> 3.19 at Keras' default epsilon 1e-7), so monotone decrease was never a property of this path.  The
asserted curve uses epsilon 1e-4 on both sides: several variables (biases in front of BatchNorm over a 2x2 map) have gradients
that are analytically zero, and Adam at epsilon 1e-7 turns their rounding noise into +-lr steps whose sign no two
implementations share.  The default-epsilon curve is printed next to the oracle's for information."""
import re

import torch


def _build(strategy, eps):
    from .core_optimizer import get_optimizer
    from .core_train import CoreTrain
    from .heads import convnext_tiny_aspp
    from .modelhelper import model_common_setup

    model = convnext_tiny_aspp(num_class=21, build_input_size=(64, 64), drop_path_rate=0.0, dropout_rate=0.0, layer_scale_init_value=1.0)
    helper = model_common_setup(model, restore_checkpoint=False)
    opt = get_optimizer(strategy, initial_lr=1e-3, optimizer="adamw", epoch_steps=10, train_epoch=1, adamw_weight_decay=0.05)
    opt.epsilon = eps
    helper.set_optimizer(opt)
    trainer = CoreTrain(helper, None).create_trainable_model(21, batch_size=2)
    return model, trainer


def _oracle_curve(model, x, y, steps, eps):
    from oracle import models as OM
    from oracle import tf_ops as O
    from .utils.train_utils import get_no_weight_decay_layers_names_from_model

    w = OM.export_weights(model)
    excl = get_no_weight_decay_layers_names_from_model(model)
    names = [p.iseg_name for p in model.parameters()]
    return OM.convnext_aspp_adamw_curve(
        w, x.cpu().double(), y.cpu(), steps, names,
        lambda s: O.warmup_poly_decay(s, 1e-3, 10, end_lr=0.0, warmup_steps=0, warmup_lr=0.0, power=0.9),
        lambda k: 0.0 if any(re.search(n, k) for n in excl) else 0.05, eps=eps)


def run_smoke(steps=4):
    from .core_env import common_env_setup
    from .data import synthetic_batch
    from oracle import models as OM

    assert torch.cuda.is_available(), "smoke() needs a GPU"
    strategy = common_env_setup(use_one_device_strategy=True, mixed_precision=False, random_seed=0)
    x, y = synthetic_batch(2, 64, 64, seed=0)
    x, y = x.cuda(), y.cuda()
    for eps, asserted in ((1e-4, True), (1e-7, False)):
        model, trainer = _build(strategy, eps)
        if asserted:      # ---- forward parity against the oracle (fp32 storage) ----
            weights = OM.export_weights(model)
            with torch.no_grad():
                logits = model(x, training=False)[0]
            ref = OM.convnext_aspp_forward(weights, x.cpu().double(), training=False)["logits"]
            err = (logits.cpu().double() - ref).abs().max().item()
            same = torch.equal(logits.argmax(-1).cpu(), ref.argmax(-1))
            print(f"smoke: forward max|logit err| = {err:.3e}, argmax identical = {same}")
            assert err < 1e-3 and same, (err, same)
        want = _oracle_curve(model, x, y, steps, eps)      # the oracle starts from the same initial weights
        got = [float(trainer.train_step(x, y)[0].detach()) for _ in range(steps)]
        torch.cuda.synchronize()
        rel = max(abs(a - b) / max(abs(b), 1e-6) for a, b in zip(got, want))
        print(f"smoke: AdamW epsilon {eps:g}: loss curve HIP {[round(v, 5) for v in got]} oracle {[round(v, 5) for v in want]} max rel diff {rel:.2e}"
              + ("" if asserted else "  (informational)"))
        if asserted:
            assert all(v == v for v in got) and rel < 1e-3, (got, want)
    print("smoke OK")
