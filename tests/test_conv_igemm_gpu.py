"""Implicit-GEMM convolution (csrc/conv_igemm.hip; reference layers/model_builder.py:54-64, layers/aspp.py:41-52,
backbones/convnext.py:72-75,255-257): forward, data gradient and weight gradient through the C ABI against the oracle's
keras Conv2D restatement over TF's "same" padding table -- kernel sizes, strides, dilations, odd / even maps, groups, split-K."""
import pytest
import torch

from oracle import tf_ops as O
from tests.test_kernels_gpu import close, q, rnd

pytestmark = pytest.mark.gpu
BF = torch.bfloat16

# (N, H, W, Cin), k, stride, dilation, Cout, groups
CASES = [
    ((2, 16, 16, 64), 3, 1, 1, 32, 1),
    ((2, 16, 16, 64), 3, 1, 6, 64, 1),       # ASPP-like: dilation larger than half the map -> most taps in the halo
    ((1, 16, 16, 768), 3, 1, 3, 256, 1),     # ASPP at the flagship size (split-K)
    ((2, 9, 7, 32), 3, 1, 2, 40, 1),         # odd map, Cout not a multiple of 16
    ((2, 9, 7, 32), 3, 2, 1, 48, 1),         # stride 2 on odd sizes
    ((2, 10, 8, 32), 3, 2, 1, 48, 1),        # stride 2 on even sizes (TF pads bottom / right only)
    ((2, 12, 12, 16), 2, 2, 1, 32, 1),       # ConvNeXt downsample 2x2 / s2
    ((2, 12, 12, 16), 2, 1, 2, 32, 1),       # its dilated form (build_dilated_convnext)
    ((1, 16, 16, 8), 4, 4, 1, 24, 1),        # patchify 4x4 / s4
    ((1, 13, 11, 8), 7, 1, 1, 16, 1),        # 7x7
    ((1, 13, 11, 8), 7, 2, 1, 16, 1),
    ((2, 8, 8, 24), 1, 2, 1, 40, 1),         # strided 1x1 (ResNet shortcut style)
    ((1, 20, 20, 16), 3, 1, 9, 16, 1),
    ((1, 24, 24, 16), 3, 1, 12, 16, 1),
    ((1, 40, 40, 8), 3, 1, 18, 8, 1),
    ((1, 5, 5, 16), 3, 4, 1, 16, 1),
    ((2, 10, 10, 32), 3, 1, 1, 64, 4),       # grouped
    ((2, 9, 9, 48), 3, 2, 2, 48, 3),         # strided AND dilated: the plain gather form
    ((2, 9, 9, 48), 3, 2, 1, 48, 3),         # grouped, strided: stride phases x groups in one launch
    ((2, 33, 31, 64), 3, 2, 1, 128, 1),      # stride phases of unequal size, several row tiles each
    ((1, 12, 12, 16), 2, 3, 1, 16, 1),       # kernel smaller than the stride: phases no tap reaches stay zero
    ((4, 32, 32, 128), 3, 1, 1, 128, 1),     # several tiles in M and a long weight-gradient reduction
    ((2, 16, 16, 128), 3, 1, 4, 192, 1),     # LDS-DMA form (Cin, Cout multiples of 64): dilated, most of the halo
    ((1, 16, 16, 768), 3, 1, 6, 256, 1),     # the flagship's ASPP branch (LDS-DMA form, split-K)
    ((3, 17, 13, 64), 3, 2, 1, 128, 1),      # LDS-DMA forward with a stride on odd sizes; data gradient by stride phase
    ((2, 12, 20, 192), 5, 1, 1, 64, 1),      # 5x5, rows not a multiple of the tile
]


def _geom(k, shape, kk, s, d, Cout, groups):
    N, H, W, C = shape
    Ho, pt = k.same_pad(H, kk, s, d)
    Wo, pl = k.same_pad(W, kk, s, d)
    return k.conv_geom(N, H, W, C, Cout, kk, kk, s, s, d, d, pt, pl, Ho, Wo, groups), Ho, Wo


@pytest.mark.parametrize("shape,kk,s,d,Cout,groups", CASES)
def test_conv_igemm_three_passes(cuda, shape, kk, s, d, Cout, groups):
    from iseg_amd import kernels as k

    N, H, W, C = shape
    geom, Ho, Wo = _geom(k, shape, kk, s, d, Cout, groups)
    assert k.conv2d_igemm_supported(geom, BF)
    x, xr = q(rnd(shape, 1), BF)
    w, wr = q(rnd((kk, kk, C // groups, Cout), 2, (kk * kk * C // groups) ** -0.5), BF)
    b = rnd((Cout,), 3).float()
    y = k.conv2d_igemm_fwd(x, w, b.cuda(), geom)
    xx, ww = xr.clone().requires_grad_(True), wr.clone().requires_grad_(True)
    yo = O.conv2d(xx, ww, b.double(), s, d, groups=groups)
    assert tuple(yo.shape) == (N, Ho, Wo, Cout) == tuple(y.shape)
    close(y, yo, BF, "igemm fwd")
    if k.conv2d_igemm_fwd_kt_supported(geom, BF):      # the LDS-DMA form on the K-contiguous kernel copy [Cout, kh*kw*Cin]
        wt = w.reshape(-1, Cout).t().contiguous()
        close(k.conv2d_igemm_fwd_kt(x, wt, b.cuda(), geom), yo, BF, "igemm fwd (LDS-DMA)")
    dy, dyr = q(rnd((N, Ho, Wo, Cout), 4), BF)
    yo.backward(dyr)
    dx = k.conv2d_igemm_bwd_data(dy, w, geom)
    close(dx, xx.grad, BF, "igemm dx", bf16_tol=1.5e-2)
    dw = torch.full((kk, kk, C // groups, Cout), 0.5, device="cuda")
    k.conv2d_igemm_bwd_weight(x, dy, dw, geom, accumulate=True)
    close(dw - 0.5, ww.grad, torch.float32, "igemm dw", f32_tol=2e-4)
    dw2 = torch.full_like(dw, 7.0)
    k.conv2d_igemm_bwd_weight(x, dy, dw2, geom, accumulate=False)
    close(dw2, ww.grad, torch.float32, "igemm dw (overwrite)", f32_tol=2e-4)


def test_conv_igemm_is_deterministic(cuda):
    """fixed-order split-K slabs and the gather-form data gradient: two runs agree bit for bit"""
    from iseg_amd import kernels as k

    shape, kk, s, d, Cout = (1, 16, 16, 768), 3, 1, 3, 256
    geom, Ho, Wo = _geom(k, shape, kk, s, d, Cout, 1)
    x, _ = q(rnd(shape, 1), BF)
    w, _ = q(rnd((kk, kk, shape[3], Cout), 2, 0.01), BF)
    dy, _ = q(rnd((1, Ho, Wo, Cout), 4), BF)
    outs = []
    for _ in range(2):
        dw = torch.zeros((kk, kk, shape[3], Cout), device="cuda")
        k.conv2d_igemm_bwd_weight(x, dy, dw, geom, accumulate=False)
        outs.append((k.conv2d_igemm_fwd(x, w, None, geom).clone(), k.conv2d_igemm_bwd_data(dy, w, geom).clone(), dw))
    for a, b_ in zip(*outs):
        assert torch.equal(a, b_)


def test_conv_igemm_refuses_what_it_cannot_do(cuda):
    from iseg_amd import _hip, kernels as k

    geom, _, _ = _geom(k, (1, 8, 8, 3), 3, 1, 1, 16, 1)      # Cin = 3: no 16-byte channel chunks
    assert not k.conv2d_igemm_supported(geom, BF)
    geom, _, _ = _geom(k, (1, 8, 8, 16), 3, 1, 1, 16, 1)
    assert not k.conv2d_igemm_supported(geom, torch.float32)
    x = torch.zeros((1, 8, 8, 16), device="cuda")
    with pytest.raises(_hip.HipCallError):
        k.conv2d_igemm_fwd(x, x, None, geom)


@pytest.mark.parametrize("dtype", [torch.float32, BF])
@pytest.mark.parametrize("groups,kk,s,d", [(1, 3, 1, 2), (4, 3, 1, 1), (2, 3, 2, 1), (1, 1, 1, 1), (2, 1, 1, 1)])
def test_conv2d_layer_routes(cuda, dtype, groups, kk, s, d):
    """keras-style Conv2D layer end to end (forward + autograd) on every route of functional._Conv2dFn: plain GEMM, implicit GEMM,
    im2col (fp32 parity mode), with and without groups"""
    from iseg_amd import nn
    from iseg_amd.layers.base_layers import Conv2D
    from iseg_amd.param_store import ParamStore
    from tests.util_models import randomize_parameters

    nn.set_compute_dtype(dtype)
    nn.set_device("cuda:0")
    try:
        shape, Cout = (2, 10, 9, 32), 48
        layer = Conv2D(Cout, kk, strides=s, padding="same", dilation_rate=d, groups=groups, use_bias=True, name="conv")
        with nn.dry_run_scope():
            layer(torch.empty(shape, dtype=dtype, device="cuda"))
        store = ParamStore(list(layer.parameters()))
        layer._iseg_store = store
        randomize_parameters(layer, 5)
        x, xr = q(rnd(shape, 1), dtype)
        xg = x.requires_grad_(True)
        y = layer(xg)
        dy, dyr = q(rnd(tuple(y.shape), 2), dtype)
        y.backward(dy)
        wr = layer.kernel.data.to(dtype).double().cpu().requires_grad_(True)
        br = layer.bias.data.double().cpu().requires_grad_(True)
        xx = xr.clone().requires_grad_(True)
        yo = O.conv2d(xx, wr, br, s, d, groups=groups)
        yo.backward(dyr)
        close(y, yo, dtype, "layer fwd", f32_tol=1e-4)
        close(xg.grad, xx.grad, dtype, "layer dx", f32_tol=1e-4, bf16_tol=1.5e-2)
        close(layer.kernel.grad, wr.grad, torch.float32, "layer dW", f32_tol=3e-4)
        close(layer.bias.grad, br.grad, torch.float32, "layer db", f32_tol=3e-4)
    finally:
        nn.set_compute_dtype(torch.float32)
