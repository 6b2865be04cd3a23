"""layers/model_builder.py of the reference on HIP operators: ConvNormAct with groups / dilation (:34-98) and reset_weights (:100-115),
SepConvBnReLU (:118-171), NormConvAct over LN / GN / BN / RMSN (:175-247), CommonEndBlock (:276-296) -- forward (train and eval) and
every gradient against the oracle, fp32 and bf16 storage."""
import pytest
import torch

from oracle import models as OM
from oracle import tf_ops as O
from tests.util_models import randomize_parameters

pytestmark = pytest.mark.gpu
DT = [torch.float32, torch.bfloat16]


def _build(layer, shape, dtype, seed=3, extra=None):
    from iseg_amd import nn
    from iseg_amd.param_store import ParamStore

    with nn.dry_run_scope():
        x = torch.empty(shape, dtype=dtype, device="cuda")
        layer([x, extra] if extra is not None else x)
    store = ParamStore(list(layer.parameters()))
    layer._iseg_store = store
    randomize_parameters(layer, seed)
    return OM.export_weights(layer)


def _bn(w, prefix, y, training, eps, relu):
    g, b = w[f"{prefix}/gamma"], w[f"{prefix}/beta"]
    if training:
        y, _, _ = O.batch_norm_train(y, g, b, eps)
    else:
        y = O.batch_norm_infer(y, g, b, w[f"{prefix}/moving_mean"], w[f"{prefix}/moving_variance"], eps)
    return torch.relu(y) if relu else y


def _check(layer, ref_fn, shape, dtype, training, extra=None, out_tol=None):
    """run layer and ref_fn(w, x) on the same rounded input, compare output, dx and every parameter gradient"""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(shape, generator=g).to(dtype)
    w = {k: v.requires_grad_(True) if v.dtype.is_floating_point else v for k, v in OM.export_weights(layer).items()}
    xg = x.cuda().requires_grad_(True)
    y = layer([xg, extra] if extra is not None else xg, training=training)
    xr = x.double().requires_grad_(True)
    yr = ref_fn(w, xr)
    assert tuple(y.shape) == tuple(yr.shape)
    dy = torch.randn(tuple(y.shape), generator=g).to(y.dtype)
    y.backward(dy.cuda())
    yr.backward(dy.double())
    bf = dtype == torch.bfloat16

    def rel(a, b):
        a, b = a.detach().cpu().double(), b.detach()
        if bf:      # a bf16-rounded pre-activation flips a few ReLUs next to zero: whole elements of the gradient differ, so compare in L2
            return (a - b).norm().item() / max(b.norm().item(), 1e-8)
        return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-8)

    assert rel(y, yr) < (out_tol or (2.5e-2 if bf else 2e-5)), rel(y, yr)
    errs = {"dx": rel(xg.grad, xr.grad)}
    for p in layer.parameters():
        if p.requires_grad and w[p.iseg_name].grad is not None:
            errs[p.iseg_name] = rel(p.grad, w[p.iseg_name].grad)
    # bf16 + batch statistics + ReLU masks in series: the fp32 rows are the parity check, the bf16 rows guard against gross errors
    bad = {k: v for k, v in errs.items() if v > (1.2e-1 if bf else 3e-4)}
    assert not bad, bad


@pytest.fixture
def compute(request):
    from iseg_amd import nn

    def set_dtype(dtype):
        nn.set_compute_dtype(dtype)
        nn.set_device("cuda:0")

    yield set_dtype
    nn.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("groups,k,dil", [(1, 3, 2), (4, 3, 1), (2, 1, 1)])
def test_conv_norm_act_groups_dilation(cuda, compute, dtype, training, groups, k, dil):
    from iseg_amd.layers.model_builder import ConvNormAct

    compute(dtype)
    layer = ConvNormAct(48, k, dilation_rate=dil, groups=groups, name="cna")
    _build(layer, (3, 9, 8, 32), dtype)

    def ref(w, x):
        y = O.conv2d(x, w["cna/conv/kernel"], None, 1, dil, "same", groups=groups)
        return _bn(w, "cna/bn", y, training, 1e-3, True)

    _check(layer, ref, (3, 9, 8, 32), dtype, training)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("training", [True, False])
def test_sep_conv_bn_relu(cuda, compute, dtype, training):
    from iseg_amd.layers.model_builder import SepConvBnReLU

    compute(dtype)
    layer = SepConvBnReLU(40, 3, dilation_rate=2, name="sep")
    _build(layer, (2, 10, 9, 24), dtype)

    def ref(w, x):
        y = O.depthwise_conv2d(x, w["sep/depthwise_conv/depthwise_kernel"], None, 1, 2)
        y = _bn(w, "sep/depthwise_bn", y, training, 1e-3, True)
        y = O.conv2d(y, w["sep/pointwise_conv/conv/kernel"], None, 1, 1, "same")
        return _bn(w, "sep/pointwise_conv/bn", y, training, 1e-3, True)

    _check(layer, ref, (2, 10, 9, 24), dtype, training)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("norm_type,groups,k", [("ln", 1, 3), ("gn", 4, 1), ("rmsn", 1, 1), ("bn", 1, 3)])
def test_norm_conv_act(cuda, compute, dtype, norm_type, groups, k):
    from iseg_amd.layers.model_builder import NormConvAct

    compute(dtype)
    layer = NormConvAct(24, k, norm_type=norm_type, groups=groups, activation="gelu", name="nca")
    _build(layer, (2, 8, 7, 16), dtype)

    def ref(w, x):
        if norm_type == "ln":
            y = O.layer_norm(x, w["nca_ln/gamma"], w["nca_ln/beta"], 1e-6)
        elif norm_type == "gn":
            y = O.group_norm(x, w["nca_ln/gamma"], w["nca_ln/beta"], groups, 1e-6)
        elif norm_type == "rmsn":
            y = O.rms_norm(x, w["nca_rmsn/scale"], 1e-6)
        else:
            y = _bn(w, "nca_bn", x, True, 1e-6, False)
        return O.gelu(O.conv2d(y, w["nca_conv/kernel"], w["nca_conv/bias"], 1, 1, "same"))

    _check(layer, ref, (2, 8, 7, 16), dtype, True)


@pytest.mark.parametrize("dtype", DT)
def test_common_end_block(cuda, compute, dtype):
    from iseg_amd.layers.model_builder import CommonEndBlock

    compute(dtype)
    layer = CommonEndBlock(32, 21, dropout_rate=0.1, name="end")
    img = torch.zeros((2, 32, 24, 3), dtype=torch.float32, device="cuda")
    _build(layer, (2, 8, 6, 16), dtype, extra=img)

    def ref(w, x):
        y = O.conv2d(x, w["end/end_conv/conv/kernel"], None, 1, 1, "same")
        y = _bn(w, "end/end_conv/bn", y, False, 1e-3, True)
        y = O.conv2d(y, w["end/logits_conv/kernel"], w["end/logits_conv/bias"], 1, 1, "same")
        return O.resize_bilinear(y, (32, 24))

    g = torch.Generator().manual_seed(1)
    x = torch.randn((2, 8, 6, 16), generator=g).to(dtype)
    y = layer([x.cuda(), img], training=False)      # eval: dropout off, moving statistics
    w = OM.export_weights(layer)
    yr = ref(w, x.double())
    assert y.dtype == torch.float32 and tuple(y.shape) == (2, 32, 24, 21)
    err = (y.cpu().double() - yr).abs().max().item() / yr.abs().max().item()
    assert err < (2.5e-2 if dtype == torch.bfloat16 else 2e-5), err


def test_reset_weights(cuda, compute):
    from iseg_amd.layers.model_builder import ConvNormAct, NormConvAct

    compute(torch.bfloat16)
    layer = ConvNormAct(32, 3, use_bias=True, name="cna")
    _build(layer, (1, 6, 6, 16), torch.bfloat16)
    before = layer.conv.kernel.data.clone()
    assert not torch.equal(layer.bn.gamma.data, torch.ones_like(layer.bn.gamma.data))
    layer.reset_weights()
    lim = (6.0 / (9 * 16 + 9 * 32)) ** 0.5      # glorot_uniform limit of a 3x3x16x32 kernel
    k = layer.conv.kernel.data
    assert not torch.equal(k, before) and k.abs().max().item() <= lim + 1e-6 and k.std().item() > 0.3 * lim
    assert torch.equal(layer.conv.bias.data, torch.zeros_like(layer.conv.bias.data))
    assert torch.equal(layer.bn.gamma.data, torch.ones_like(layer.bn.gamma.data))
    assert torch.equal(layer.bn.beta.data, torch.zeros_like(layer.bn.beta.data))
    assert torch.equal(layer.bn.moving_mean, torch.zeros_like(layer.bn.moving_mean))
    assert torch.equal(layer.bn.moving_variance, torch.ones_like(layer.bn.moving_variance))
    # the bf16 copies the kernels read follow the masters
    assert torch.equal(layer.conv.kernel.iseg_compute, k.to(torch.bfloat16))
    first = k.clone()
    layer.reset_weights()
    assert not torch.equal(layer.conv.kernel.data, first)      # a fresh draw each time
    nca = NormConvAct(8, 1, name="nca")
    _build(nca, (1, 4, 4, 16), torch.bfloat16)
    nca.reset_weights()
    assert torch.equal(nca.ln.gamma.data, torch.ones_like(nca.ln.gamma.data)) and torch.equal(nca.conv.bias.data, torch.zeros_like(nca.conv.bias.data))
