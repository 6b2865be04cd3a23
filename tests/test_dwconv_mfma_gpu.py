"""Depthwise 7 x 7 on the matrix cores (csrc/dwconv_mfma.hip, round 5): keras DepthwiseConv2D(7, padding="same") of the ConvNeXt block
(backbones/convnext.py:23-27, 47-50) and its data gradient, bf16 storage, against the fp64 oracle on the same bf16-rounded operands.  The kernel rounds
its weights to bf16 (the reference's mixed_bfloat16 policy does the same); handing it bf16-representable weights makes every product exact, so only the
fp32 summation order and the final bf16 rounding separate the two sides."""
import pytest
import torch

from oracle import tf_ops as O

pytestmark = pytest.mark.gpu


def _rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float64) * scale


@pytest.mark.parametrize("N,H,W,C", [(2, 16, 16, 32), (1, 33, 31, 96), (2, 17, 48, 64), (1, 5, 7, 32), (1, 64, 64, 192), (3, 32, 32, 384)])
def test_dwconv7_mfma_forward_matches_oracle(cuda, N, H, W, C):
    from iseg_amd import kernels as K

    bf = torch.bfloat16
    x = _rnd((N, H, W, C), 1).to(bf)
    w = (_rnd((7, 7, C, 1), 2) / 7).to(bf).double()      # bf16-representable weights
    b = _rnd((C,), 3, 0.3).float()
    y = K.dwconv2d7_mfma(x.cuda(), w.reshape(49, C).float().cuda(), b.cuda())
    ref = O.depthwise_conv2d(x.double(), w, b.double(), 1, 1, "same")
    err = (y.cpu().double() - ref).abs()
    assert (err <= 2.0 ** -8 * ref.abs() + 1e-5).all(), (err.max().item(), ref.abs().max().item())


@pytest.mark.parametrize("N,H,W,C", [(2, 16, 16, 32), (1, 33, 31, 96), (2, 40, 24, 64)])
def test_dwconv7_mfma_data_gradient_with_residual_matches_oracle(cuda, N, H, W, C):
    """flip = 1 + add: dx = conv(dy, flipped kernel) + the gradient arriving through the residual branch, rounded ONCE"""
    from iseg_amd import kernels as K

    bf = torch.bfloat16
    dy = _rnd((N, H, W, C), 4).to(bf)
    res = _rnd((N, H, W, C), 5).to(bf)
    w = (_rnd((7, 7, C, 1), 6) / 7).to(bf).double()
    dx = K.dwconv2d7_mfma(dy.cuda(), w.reshape(49, C).float().cuda(), None, flip=True, add=res.cuda())
    xr = torch.zeros((N, H, W, C), dtype=torch.float64, requires_grad=True)
    O.depthwise_conv2d(xr, w, None, 1, 1, "same").backward(dy.double())
    ref = xr.grad + res.double()
    err = (dx.cpu().double() - ref).abs()
    assert (err <= 2.0 ** -8 * ref.abs() + 1e-5).all(), (err.max().item(), ref.abs().max().item())


def test_dwconv7_mfma_is_bit_reproducible_and_confines_non_finite_inputs(cuda):
    """two launches give the same bits.  A non-finite pixel reaches the outputs whose 7 x 7 window contains it -- and, a property of the banded
    product (the matrix pipe multiplies the pixel by the band's zeros as well: 0 x inf = NaN), up to 5 more output rows of the same 12-row band in
    those 7 columns of that channel; nothing else.  (The VALU kernels touch the 49 outputs only; a training step with a non-finite activation is
    lost either way.)"""
    from iseg_amd import kernels as K

    bf = torch.bfloat16
    x = _rnd((1, 40, 40, 32), 7).to(bf)
    x[0, 30, 10, 5] = float("inf")
    w = (_rnd((49, 32), 8) / 7).float()
    a = K.dwconv2d7_mfma(x.cuda(), w.cuda(), None)
    b = K.dwconv2d7_mfma(x.cuda(), w.cuda(), None)
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    bad = ~torch.isfinite(a.float().cpu())
    assert bad[0, 27:34, 7:14, 5].all()
    allowed = torch.zeros_like(bad)
    allowed[0, 22:38, 7:14, 5] = True
    assert not (bad & ~allowed).any() and int(bad.sum()) <= 7 * 12


def test_dwconv7_mfma_rejects_other_shapes_and_the_automatic_route_agrees(cuda):
    from iseg_amd import _hip, kernels as K

    bf = torch.bfloat16
    with pytest.raises(_hip.HipCallError):
        K.dwconv2d7_mfma(torch.zeros(1, 8, 8, 24, dtype=bf, device="cuda"), torch.zeros(49, 24, device="cuda"), None)      # C % 32 != 0
    # a plane large enough for iseg_dwconv2d_fwd to pick the matrix-core route by itself (>= 2048 units): identical bits to the named entry
    x = _rnd((16, 128, 128, 64), 9).to(bf).cuda()      # 16 x 64 tiles x 2 slabs = 2 048 units
    w = (_rnd((49, 64), 10) / 7).float().cuda()
    b = _rnd((64,), 11, 0.3).float().cuda()
    assert torch.equal(K.dwconv2d(x, w, b, 7, 1, 3, 3).view(torch.int16), K.dwconv2d7_mfma(x, w, b).view(torch.int16))
