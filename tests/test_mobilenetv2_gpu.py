"""MobileNetV2 (reference backbones/mobilenetv2_common.py) through get_backbone against the fp64 restatement, which keeps the reference's
ZeroPadding2D + 'valid' form of the stride-2 blocks: endpoints at several output strides and sizes (even and odd), and a training-mode
gradient check of the whole network in fp32."""
import pytest
import torch

from oracle import models as OM
from tests.util_models import randomize_parameters

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return (a.detach().cpu().double() - b).abs().max().item() / max(b.abs().max().item(), 1e-8)


@pytest.mark.parametrize("output_stride,size", [(32, (64, 96)), (16, (65, 47)), (8, (64, 64))])
def test_mobilenetv2_endpoints_match_oracle(cuda, output_stride, size):
    from iseg_amd import nn
    from iseg_amd.backbones.feature_extractor import get_backbone
    from iseg_amd.param_store import ParamStore

    nn.set_compute_dtype(torch.float32)
    nn.set_device("cuda:0")
    m = get_backbone("mobilenetv2", output_stride=output_stride, return_endpoints=True, image_shape=(1, 64, 64, 3))
    m._iseg_store = ParamStore(list(m.parameters()))
    randomize_parameters(m, 3)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, size[0], size[1], 3, generator=g)
    with torch.no_grad():
        ends = m(x.cuda(), training=False)
    ref = OM.mobilenetv2_forward(OM.export_weights(m), x.double(), output_stride)
    assert len(ends) == len(ref) == 5
    for got, want in zip(ends, ref):
        assert tuple(got.shape) == tuple(want.shape)
        assert _rel(got, want) < 2e-4


def test_mobilenetv2_training_gradients_fp32(cuda):
    from iseg_amd import nn
    from iseg_amd.backbones.feature_extractor import get_backbone
    from iseg_amd.param_store import ParamStore

    nn.set_compute_dtype(torch.float32)
    nn.set_device("cuda:0")
    m = get_backbone("mobilenetv2", output_stride=16, return_endpoints=False, image_shape=(1, 64, 64, 3))
    m._iseg_store = ParamStore(list(m.parameters()))
    randomize_parameters(m, 8)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 64, 64, 3, generator=g)
    y = m(x.cuda(), training=True)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.cuda())
    w = {k: v.requires_grad_(True) if v.is_floating_point() else v for k, v in OM.export_weights(m).items()}
    ref = OM.mobilenetv2_forward(w, x.double(), 16, training=True)[-1]
    ref.backward(dy.double())
    assert _rel(y, ref.detach()) < 5e-4
    # Per tensor, the L2 error against the larger of the tensor's own norm and a small fraction of the largest gradient norm (a beta in front of
    # a 1x1 convolution + training-mode BN has an exactly zero gradient).  L2, not the largest element: a relu6 gate whose pre-activation
    # sits within rounding of 0 or 6 (fp32 statistics here, fp64 in the oracle) switches single elements on or off, which moves the largest
    # element by per cent while the tensor as a whole stays put.
    gmax = max(w[p.iseg_name].grad.norm().item() for p in m.parameters())
    errs = {}
    for p in m.parameters():
        ref_g = w[p.iseg_name].grad
        errs[p.iseg_name] = (p.grad.cpu().double() - ref_g).norm().item() / max(ref_g.norm().item(), 1e-3 * gmax)
    bad = {k: round(v, 5) for k, v in errs.items() if v > 5e-2}
    assert not bad, bad
