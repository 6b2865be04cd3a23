"""MOAT (backbones/moat/* of the reference): a reduced member of the family -- stem, MBConv stages with squeeze-and-excitation, MOAT stages (MBConv
without SE + global self-attention over the stage's tokens), strided depthwise convolutions, average-pooled shortcuts, drop path with injected
factors -- against the oracle's restatement, forward (every endpoint) and every gradient; plus the registry names."""
import pytest
import torch

from oracle import models as OM
from tests.test_attention_gpu import _check_grads, _rel, _setup
from tests.test_kernels_gpu import DTYPES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("training,pos", [(False, None), (True, None), (False, [None, None, 4, 2]), (True, [None, None, 4, 2])])
def test_moat_reduced_family_member(cuda, dtype, training, pos):
    """pos: position_embedding_size per stage -- with it the MOAT blocks carry the 2-D relative position embedding (tables [heads, 7, 7] and [heads, 3, 3]
    resized bilinearly to [11, 15] and [5, 7], re-indexed to one bias per head; a build-time constant in the reference: the tables get no gradient)"""
    from iseg_amd import nn
    from iseg_amd.backbones.moat.moat import MOAT

    nn.set_compute_dtype(dtype)
    nn.set_device("cuda:0")
    try:
        kinds, blocks = ["mbconv", "mbconv", "moat", "moat"], [1, 2, 2, 1]
        moat = MOAT(stem_size=[16, 16], block_type_list=kinds, num_blocks=blocks, hidden_size=[16, 32, 64, 96], head_size=32,
                    position_embedding_size=pos, survival_prob=0.8, return_endpoints=True, name="moat")
        shape = (2, 96, 128, 3)
        _setup(moat, torch.empty(shape, dtype=torch.float32, device="cuda"))
        f1, f2 = torch.tensor([1.25, 0.0]), torch.tensor([0.0, 1.25])
        dp = {}
        if training:
            moat._blocks[1][1].drop_path_mask = f1.cuda()
            moat._blocks[2][0].drop_path_masks = (f2.cuda(), f1.cuda())
            dp = {"moat/block_01_01": f1.double(), "moat/block_02_00": (f2.double(), f1.double())}
            for st in moat._blocks:      # every other stochastic depth draw: off (survival 1 keeps the arithmetic, the RNG stream cannot match TF's)
                for blk in st:
                    if getattr(blk, "drop_path_mask", 0) is None and not hasattr(blk, "drop_path_masks"):
                        blk.survival_prob = None
                    if hasattr(blk, "drop_path_masks") and blk.drop_path_masks is None:
                        blk.survival_prob = None
        g = torch.Generator().manual_seed(0)
        x = torch.randn(shape, generator=g)
        ends = moat(x.cuda(), training=training)
        w = {k_: (v.requires_grad_(True) if not k_.endswith(("moving_mean", "moving_variance")) else v) for k_, v in OM.export_weights(moat).items()}
        ref = OM.moat_forward(w, x.double(), "moat", kinds, blocks, training=training, dp_factors=dp)
        assert [tuple(e.shape) for e in ends] == [tuple(r.shape) for r in ref] == [(2, 48, 64, 16), (2, 24, 32, 16), (2, 12, 16, 32), (2, 6, 8, 64),
                                                                                     (2, 3, 4, 96)]
        tol = 2e-4 if dtype == torch.float32 else 6e-2
        for i, (a, b) in enumerate(zip(ends, ref)):
            assert _rel(a, b.detach()) < tol, f"endpoint {i}"
        dy = torch.randn(tuple(ref[-1].shape), generator=g)
        ends[-1].backward(dy.cuda().to(dtype))
        ref[-1].backward(dy.to(dtype).double())
        # batch statistics over 2 x 3 x 4 positions at the last stage: an fp32 forward differs from fp64 by 1e-6, BatchNorm's backward amplifies it
        # (with batch statistics a constant added in front of a BatchNorm-normalised branch has an analytically ZERO gradient -- the stem's first bias,
        # every pre_norm beta (a per-channel constant through the bias-free expand convolution into expand_norm): only rounding noise is left to compare)
        skip = ("pre_norm/beta", "stem/conv_0/bias") if training else ()
        _check_grads(moat, w, (2e-3 if training else 5e-4) if dtype == torch.float32 else 0.15, l2=True, skip=skip)
        tables = [p for p in moat.parameters() if p.iseg_name.endswith("relative_position_embedding")]
        assert len(tables) == (3 if pos else 0)
        assert [tuple(p.shape) for p in tables] == ([(2, 7, 7), (2, 7, 7), (3, 3, 3)] if pos else [])
        assert all(p.grad is None or float(p.grad.abs().max()) == 0.0 for p in tables)      # (the reference's bias is a constant of build())
    finally:
        nn.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("pos", [None, [None, None, 3, 2]])
def test_moat_windowed_attention(cuda, dtype, pos):
    """MOATBlock(window_size=[h, w]) (backbones/moat/moat_blocks.py:317-327, 407-434, 486-497 -- round-5 verdict, missing item 2): the stage's map is cut
    into non-overlapping windows, the attention (and its relative position bias, sized by the WINDOW) runs inside each, the windows are put back;
    forward (every endpoint) and every gradient against the oracle; a map that is not a whole number of windows is an error as in the reference's
    reshape"""
    from iseg_amd import nn
    from iseg_amd.backbones.moat.moat import MOAT

    nn.set_compute_dtype(dtype)
    nn.set_device("cuda:0")
    try:
        kinds, blocks = ["mbconv", "mbconv", "moat", "moat"], [1, 1, 2, 1]
        windows = [None, None, [3, 4], [3, 2]]      # stage 2 map 6 x 8 -> four 3 x 4 windows per image; stage 3 map 3 x 4 -> two 3 x 2 windows
        moat = MOAT(stem_size=[16, 16], block_type_list=kinds, num_blocks=blocks, hidden_size=[16, 32, 64, 96], head_size=32,
                    position_embedding_size=pos, window_size=windows, survival_prob=None, return_endpoints=True, name="moat")
        shape = (2, 96, 128, 3)
        _setup(moat, torch.empty(shape, dtype=torch.float32, device="cuda"))
        g = torch.Generator().manual_seed(0)
        x = torch.randn(shape, generator=g)
        ends = moat(x.cuda(), training=False)
        w = {k_: (v.requires_grad_(True) if not k_.endswith(("moving_mean", "moving_variance")) else v) for k_, v in OM.export_weights(moat).items()}
        ref = OM.moat_forward(w, x.double(), "moat", kinds, blocks, training=False, window_size=windows)
        tol = 2e-4 if dtype == torch.float32 else 6e-2
        for i, (a, b) in enumerate(zip(ends, ref)):
            assert tuple(a.shape) == tuple(b.shape) and _rel(a, b.detach()) < tol, f"endpoint {i}"
        # the windows matter: the same weights without them give another answer at the MOAT stages
        plain = OM.moat_forward(w, x.double(), "moat", kinds, blocks, training=False) if (pos is None and dtype == torch.float32) else None
        if plain is not None:
            assert _rel(ends[3], plain[3].detach()) > 10 * tol
        dy = torch.randn(tuple(ref[-1].shape), generator=g)
        ends[-1].backward(dy.cuda().to(dtype))
        ref[-1].backward(dy.to(dtype).double())
        _check_grads(moat, w, 5e-4 if dtype == torch.float32 else 0.15, l2=True)
        if pos is not None:      # the bias tables are sized / resized by the window (scale ratio = window / position_embedding_size, :380-392)
            tables = [p for p in moat.parameters() if p.iseg_name.endswith("relative_position_embedding")]
            assert [tuple(p.shape) for p in tables] == [(2, 5, 5), (2, 5, 5), (3, 3, 3)]
        bad = MOAT(stem_size=[16, 16], block_type_list=kinds, num_blocks=blocks, hidden_size=[16, 32, 64, 96], head_size=32, position_embedding_size=None,
                   window_size=[None, None, [4, 4], None], survival_prob=None, return_endpoints=True, name="moat_bad")
        with pytest.raises(ValueError):      # 6 x 8 map, 4 x 4 windows (already the shape-only build pass trips over it)
            _setup(bad, torch.empty(shape, dtype=torch.float32, device="cuda"))
            bad(x.cuda(), training=False)
    finally:
        nn.set_compute_dtype(torch.float32)


def test_moat_names_are_registered_with_and_without_position_embedding(cuda):
    from iseg_amd import nn
    from iseg_amd import static_strings as ss
    from iseg_amd.backbones.feature_extractor import get_backbone

    nn.set_device("cuda:0")
    b = get_backbone(ss.MOAT0, return_endpoints=True, image_shape=(1, 128, 128, 3))      # (moat_use_pos_encoding=False: the reference's default)
    with nn.dry_run_scope():
        ends = b(torch.empty((1, 128, 128, 3), device="cuda"))
    assert [tuple(e.shape) for e in ends] == [(1, 64, 64, 64), (1, 32, 32, 96), (1, 16, 16, 192), (1, 8, 8, 384), (1, 4, 4, 768)]
    b = get_backbone(ss.MOAT0, return_endpoints=True, image_shape=(1, 224, 224, 3), moat_use_pos_encoding=True)
    with nn.dry_run_scope():
        ends = b(torch.empty((1, 224, 224, 3), device="cuda"))
    assert tuple(ends[-1].shape) == (1, 7, 7, 768)
    tables = sorted(tuple(p.shape) for p in b.parameters() if p.iseg_name.endswith("relative_position_embedding"))
    assert tables == sorted([(12, 27, 27)] * 7 + [(24, 13, 13)] * 2)      # stride 16: P = 14 -> 27 x 27 per head; stride 32: P = 7 -> 13 x 13
