"""shared helpers for the model-level parity tests"""
import math

import torch


def randomize_parameters(model, seed=0):
    """kernels ~ N(0, fan_in^-0.5), biases / beta ~ N(0, 0.1), gamma / layer-scale ~ U(0.5, 1.5),
    BN moving mean ~ N(0, 0.1), moving variance ~ U(0.5, 1.5) -- NOT the 1e-6 layer-scale default, which would hide
    block errors."""
    g = torch.Generator().manual_seed(seed)
    for p in model.parameters():
        name = p.iseg_name.split("/")[-1]
        shape = tuple(p.shape)
        if name in ("kernel", "depthwise_kernel"):
            if name == "depthwise_kernel":
                fan_in = shape[0] * shape[1]
            elif len(shape) == 4:
                fan_in = shape[0] * shape[1] * shape[2]
            else:
                fan_in = shape[0]
            v = torch.randn(shape, generator=g) * fan_in ** -0.5
        elif name in ("bias", "beta"):
            v = torch.randn(shape, generator=g) * 0.1
        elif name == "gamma":
            v = torch.rand(shape, generator=g) + 0.5
        else:
            v = torch.randn(shape, generator=g) * 0.02
        p.data.copy_(v.to(p.device))
    for b in model.buffers():
        name = getattr(b, "iseg_name", "").split("/")[-1]
        if name == "moving_mean":
            b.copy_((torch.randn(tuple(b.shape), generator=g) * 0.1).to(b.device))
        elif name == "moving_variance":
            b.copy_((torch.rand(tuple(b.shape), generator=g) + 0.5).to(b.device))
    store = getattr(model, "_iseg_store", None)
    if store is not None:
        store.sync_shadow()
