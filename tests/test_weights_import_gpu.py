"""Pretrained-weight import on the device (reference saver/h5_saver.py:38-298, backbones/feature_extractor.py:166-187): a
backbone file in the Keras group structure is loaded through get_backbone(name, weights_path=...) / SegManaged(backbone_weights_path=...)
onto cuda:0, by exact names and through the fuzzy match of a renamed file, and the device model then reproduces the oracle's forward with
THOSE weights -- fp32 storage to 1e-3 with a bit-exact argmax mask, bf16 storage (shadows refreshed from the loaded masters) to bf16
rounding."""
import numpy as np
import pytest
import torch

from oracle import models as OM
from oracle import tf_ops as O
from tests.util_models import randomize_parameters

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_policy():
    from iseg_amd import nn

    yield
    nn.set_compute_dtype(torch.float32)


def _donor(tmp_path):
    """a ConvNeXt-T with SURVEY 8(d) weights, written as a Keras-style weight file (one group per direct layer) and as a renamed variant
    ('.'-separated layer GROUP names, as Keras 3 writes them: the replace_slash route of saver/h5_saver.py:58-61.  The weight names stay exact:
    this backbone numbers its blocks ("stages/1/1/gamma"), and the reference's component-PRESENCE match (:281-298) cannot tell
    "stages/1/0/gamma" from "stages/1/1/gamma" -- its fuzzy route is exercised on distinctly named layers in tests/test_saver.py)"""
    from iseg_amd import nn
    from iseg_amd.backbones.feature_extractor import get_backbone
    from iseg_amd.saver import open_weights, save_weights, write_npz

    nn.set_device("cuda:0")
    donor = get_backbone("convnext_tiny", image_shape=(1, 64, 64, 3), return_endpoints=True)
    randomize_parameters(donor, 31)
    exact = save_weights(donor, str(tmp_path / "convnext_tiny.h5.npz"))
    root = open_weights(exact)
    layers = {}
    for lname in [str(s) for s in root.attrs["layer_names"]]:
        g = root[lname]
        layers[lname.replace("/", ".")] = {w: np.asarray(g[w]) for w in [str(s) for s in g.attrs["weight_names"]]}
    renamed = write_npz(str(tmp_path / "renamed.h5.npz"), layers)
    return donor, exact, renamed


@pytest.mark.parametrize("which", ["exact", "renamed"])
def test_get_backbone_weights_path_on_the_device_reproduces_the_oracle(cuda, tmp_path, which):
    from iseg_amd import nn
    from iseg_amd.backbones.feature_extractor import get_backbone
    from iseg_amd.data import synthetic_batch

    nn.set_compute_dtype(torch.float32)
    donor, exact, renamed = _donor(tmp_path)
    nn.set_seed(99)      # another initialisation: everything must come from the file
    bb = get_backbone("convnext_tiny", image_shape=(1, 96, 128, 3), return_endpoints=True, weights_path=exact if which == "exact" else renamed)
    assert all(p.is_cuda for p in bb.parameters())
    want = {p.iseg_name: p for p in donor.parameters()}
    for p in bb.parameters():
        assert torch.equal(p.data, want[p.iseg_name].data), p.iseg_name
    x, _ = synthetic_batch(2, 96, 128, seed=8)
    with torch.no_grad():
        ends = bb(x.cuda(), training=False)
    ref = OM.convnext_backbone(OM.export_weights(donor), x.double())
    assert ends[0] is None and ref[0] is None
    for a, b in zip(ends[1:], ref[1:]):
        assert (a.cpu().double() - b).abs().max().item() < 1e-3 * max(1.0, b.abs().max().item())


def test_segmanaged_backbone_weights_path_fp32_and_bf16(cuda, tmp_path):
    from iseg_amd import nn
    from iseg_amd.data import synthetic_batch
    from iseg_amd.heads import ASPPHead
    from iseg_amd.layers.core_model_ext import SegManaged
    from iseg_amd.param_store import ParamStore

    donor, exact, _ = _donor(tmp_path)
    x, _ = synthetic_batch(2, 64, 96, seed=12)
    logits = {}
    for dtype in (torch.float32, torch.bfloat16):
        nn.set_compute_dtype(dtype)
        nn.set_seed(7)
        model = SegManaged(backbone_name="convnext_tiny", backbone_weights_path=exact, output_stride=32, num_class=21, build_input_size=(64, 96),
                           name="seg")
        model.head = ASPPHead(256, output_stride=32, dropout_rate=0.0)
        model.build_with_dummy()
        # the head keeps its own (seeded) initialisation; the backbone must carry the file's weights
        donor_w = {p.iseg_name: p.data for p in donor.parameters()}
        loaded = [p for p in model.parameters() if p.iseg_name in donor_w]
        assert len(loaded) == len(donor_w)
        for p in loaded:
            assert torch.equal(p.data, donor_w[p.iseg_name]), p.iseg_name
        model._iseg_store = ParamStore(list(model.parameters()))      # (bf16: builds the compute shadows from the loaded masters)
        with torch.no_grad():
            logits[dtype] = model(x.cuda(), training=False)[0].float().cpu()
        if dtype == torch.float32:
            ref = OM.convnext_aspp_forward(OM.export_weights(model), x.double(), training=False)["logits"]
            assert (logits[dtype].double() - ref).abs().max().item() < 1e-3
            assert torch.equal(logits[dtype].argmax(-1), O.argmax_first(ref))
    d = (logits[torch.bfloat16] - logits[torch.float32]).abs().max().item()
    assert d < 0.08 * max(1.0, logits[torch.float32].abs().max().item()), d
    agree = (logits[torch.bfloat16].argmax(-1) == logits[torch.float32].argmax(-1)).float().mean().item()
    assert agree > 0.97, agree
