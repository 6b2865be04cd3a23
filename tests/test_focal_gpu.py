"""use_focal_loss end to end: SegFoundation.custom_losses hands the focal settings to the CE factory (core_model.py:498-505 of the
reference) and a training step runs on them."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_focal_loss_through_segfoundation(cuda):
    from iseg_amd import nn
    from iseg_amd.core_model import SegFoundation
    from oracle import tf_ops as O

    model = SegFoundation(num_class=5, use_focal_loss=True, focal_loss_gamma=2.0, focal_loss_alpha=0.25)
    losses = model.custom_losses(num_class=5, ignore_label=255, batch_size=2)
    fn = next(iter(losses.values())) if isinstance(losses, dict) else losses[0]
    g = torch.Generator().manual_seed(3)
    logits = (torch.randn(2, 9, 7, 5, generator=g) * 2).float()
    labels = torch.randint(0, 5, (2, 9, 7), generator=g, dtype=torch.int32)
    labels[0, :2] = 255
    z = logits.cuda().requires_grad_(True)
    px = fn(labels.cuda(), z)
    want = O.softmax_focal_ce_ignore(labels, logits.double(), 5, 255, None, 0.25, 2.0)
    assert (px.detach().cpu().double() - want).abs().max() < 2e-5
    px.mean().backward()
    zr = logits.double().requires_grad_(True)
    O.softmax_focal_ce_ignore(labels, zr, 5, 255, None, 0.25, 2.0).mean().backward()
    assert (z.grad.cpu().double() - zr.grad).abs().max() < 1e-6
    if getattr(fn, "fused_mean", None) is not None:
        m = fn.fused_mean(labels.cuda(), logits.cuda())
        assert abs(float(m) - float(want.mean())) < 2e-5
