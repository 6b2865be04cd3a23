"""EVA-02 (backbones/eva/* of the reference): the rotary-embedding and gated-product kernels of csrc/eva.hip through the C ABI, the three
feed-forward variants, EvaAttention in both projection layouts, EvaBlock's norm / layer-scale / drop-path variants and a reduced EVA02 trunk with
position-embedding resampling -- all against the oracle's line-by-line restatement (oracle/models.py eva_*), forward and every gradient."""
import pytest
import torch

from oracle import models as OM
from oracle import tf_ops as O
from tests.test_attention_gpu import _check_grads, _rel, _setup
from tests.test_kernels_gpu import DTYPES, rnd

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,heads,hd,prefix", [(2, 3, 5, 2, 64, 1), (1, 4, 4, 3, 32, 0), (3, 2, 7, 1, 8, 1)])
def test_qkv_rope_kernel_matches_the_reference_formula_and_its_inverse_is_the_transpose(cuda, dtype, B, H, W, heads, hd, prefix):
    """iseg_qkv_rope: bias [q_bias | 0 | v_bias] + apply_rot_embed_cat on q, k of the non-prefix tokens (attention.py:100-112,136-146)"""
    from iseg_amd import kernels as K
    from iseg_amd import nn
    from iseg_amd.backbones.eva.rotar_embedding_cat import RotaryEmbeddingCat

    nn.set_device("cuda:0")
    C, T = heads * hd, prefix + H * W
    qkv = rnd((B, T, 3 * C), 1).to(dtype)
    qb, vb = rnd((C,), 2, 0.3).float(), rnd((C,), 3, 0.3).float()
    emb = RotaryEmbeddingCat(filters=hd, in_pixels=False)([H, W])
    sin, cos = OM.eva_rope_table(H, W, hd)
    # (the product builds its table with torch, the oracle with numpy: float32 sin / cos / pow agree to an ulp or two)
    assert (emb.cpu().double() - torch.cat([sin, cos], dim=-1)).abs().max().item() < 2e-6, "the product's table and the oracle's differ"
    got = K.qkv_rope(qkv.cuda().clone(), qb.cuda(), vb.cuda(), emb, T, prefix, C, hd).cpu().double()
    x = qkv.double() + torch.cat([qb.double(), torch.zeros(C, dtype=torch.float64), vb.double()])
    parts = []
    for i in range(3):
        p = x[..., i * C:(i + 1) * C].reshape(B, T, heads, hd).permute(0, 2, 1, 3)
        if i < 2:
            p = torch.cat([p[:, :, :prefix], p[:, :, prefix:] * cos + OM.eva_rot(p[:, :, prefix:]) * sin], dim=2)
        parts.append(p.permute(0, 2, 1, 3).reshape(B, T, C))
    ref = torch.cat(parts, dim=-1)
    tol = 1e-5 if dtype == torch.float32 else 2.0 ** -7
    assert (got - ref).abs().max().item() <= tol * ref.abs().max().item()
    if dtype == torch.float32:      # <R a, b> == <a, R^T b>
        a, b = rnd((B, T, 3 * C), 4).float().cuda(), rnd((B, T, 3 * C), 5).float().cuda()
        Ra = K.qkv_rope(a.clone(), None, None, emb, T, prefix, C, hd)
        Rtb = K.qkv_rope(b.clone(), None, None, emb, T, prefix, C, hd, inverse=True)
        lhs, rhs = (Ra.double() * b.double()).sum().item(), (a.double() * Rtb.double()).sum().item()
        assert abs(lhs - rhs) < 1e-5 * max(1.0, abs(lhs))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("activation", ["gelu", "swish", "sigmoid"])
@pytest.mark.parametrize("packed,width", [(0, 8), (1, 8), (2, 8), (0, 2), (0, 1)])      # width: 8 = 16-byte pieces; 2 / 1 = the scalar form (2730 / 263 columns)
def test_glu_forward_and_gradients(cuda, dtype, activation, packed, width):
    """F.glu / F.glu_packed (csrc/eva.hip iseg_glu_fwd / _bwd): act(gate) * x on separate and on packed operands, both gate positions"""
    from iseg_amd import functional as F
    from iseg_amd import nn

    nn.set_compute_dtype(dtype)
    nn.set_device("cuda:0")
    try:
        act = OM._eva_act(activation)
        M, Hd = (37, 48) if width == 8 else (41, 2730 if width == 2 else 263)
        dy = rnd((M, Hd), 3).to(dtype)
        if packed:
            p = rnd((M, 2 * Hd), 1).to(dtype)
            pg = p.cuda().requires_grad_(True)
            y = F.glu_packed(pg, activation, gate_last=packed == 1)
            pr = p.double().requires_grad_(True)
            x1, x2 = pr[:, :Hd], pr[:, Hd:]
            yr = x1 * act(x2) if packed == 1 else act(x1) * x2
            y.backward(dy.cuda())
            yr.backward(dy.double())
            pairs = [(pg.grad, pr.grad)]
        else:
            g, x = rnd((M, Hd), 1).to(dtype), rnd((M, Hd), 2).to(dtype)
            gg, xg = g.cuda().requires_grad_(True), x.cuda().requires_grad_(True)
            y = F.glu(gg, xg, activation)
            gr, xr = g.double().requires_grad_(True), x.double().requires_grad_(True)
            yr = act(gr) * xr
            y.backward(dy.cuda())
            yr.backward(dy.double())
            pairs = [(gg.grad, gr.grad), (xg.grad, xr.grad)]
        tol = 2e-6 if dtype == torch.float32 else 1.2e-2
        assert (y.detach().cpu().double() - yr.detach()).abs().max().item() <= tol * yr.abs().max().item()
        for got, want in pairs:
            assert (got.cpu().double() - want).abs().max().item() <= tol * want.abs().max().item()
    finally:
        nn.set_compute_dtype(torch.float32)


_BLOCKS = [
    # (constructor keywords, oracle keywords): EVA02-tiny style, EVA02-large style, plain Mlp with sub-LN + layer scale + post norm
    (dict(num_heads=1, qkv_fused=True, mlp_ratio=8 / 3, swiglu_mlp=True, scale_mlp=False), dict(fused=True, mlp_kind="glu")),
    (dict(num_heads=2, qkv_fused=False, mlp_ratio=8 / 3, swiglu_mlp=True, scale_mlp=True, scale_attention_inner=True),
     dict(fused=False, mlp_kind="swiglu", scale_attention_inner=True)),
    (dict(num_heads=4, qkv_fused=True, mlp_ratio=2.0, swiglu_mlp=False, scale_mlp=True, init_values=0.7, use_post_norm=True),
     dict(fused=True, mlp_kind="mlp_norm", post_norm=True)),
    # hidden widths that are not multiples of 8 (EVA02-large: int(1024 * 8 / 3) = 2730): 96 * 2.73 -> 262 (rows 4-byte aligned), 96 * 2.74 -> 263 (2-byte):
    # scalar GLU, one-wavefront-per-row LayerNorm over the hidden units, register-staged GEMMs
    (dict(num_heads=2, qkv_fused=False, mlp_ratio=2.73, swiglu_mlp=True, scale_mlp=True, scale_attention_inner=True),
     dict(fused=False, mlp_kind="swiglu", scale_attention_inner=True)),
    (dict(num_heads=2, qkv_fused=False, mlp_ratio=2.74, swiglu_mlp=True, scale_mlp=True, scale_attention_inner=True),
     dict(fused=False, mlp_kind="swiglu", scale_attention_inner=True)),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("variant", range(len(_BLOCKS)))
def test_eva_block_variants(cuda, dtype, variant):
    from iseg_amd import nn
    from iseg_amd.backbones.eva.block import EvaBlock
    from iseg_amd.backbones.eva.rotar_embedding_cat import RotaryEmbeddingCat

    kw, okw = _BLOCKS[variant]
    nn.set_compute_dtype(dtype)
    nn.set_device("cuda:0")
    try:
        C, H, W, B = (96, 192, 96, 96, 96)[variant], 3, 5, 2
        heads = kw["num_heads"]
        blk = EvaBlock(drop_path_rate=0.2, class_token_size=1, name="blk", **kw)
        x = rnd((B, 1 + H * W, C), 1).to(dtype)
        _setup(blk, torch.empty(tuple(x.shape), dtype=dtype, device="cuda"))
        f = torch.tensor([1.25, 0.0])
        blk.drop_path_mask = f.cuda()
        rope = RotaryEmbeddingCat(filters=C // heads, in_pixels=False)([H, W])
        xg = x.cuda().requires_grad_(True)
        y = blk(xg, rope=rope, training=True)
        w = {k_: v.requires_grad_(True) for k_, v in OM.export_weights(blk).items()}
        xr = x.double().requires_grad_(True)
        yr = OM.eva_block(w, "blk", xr, heads, rope=OM.eva_rope_table(H, W, C // heads), prefix=1, dp=f.double(), **okw)
        assert _rel(y, yr.detach()) < (2e-5 if dtype == torch.float32 else 4e-2)
        dy = rnd(tuple(yr.shape), 7).to(dtype)
        y.backward(dy.cuda())
        yr.backward(dy.double())
        assert _rel(xg.grad, xr.grad) < (2e-4 if dtype == torch.float32 else 6e-2)
        _check_grads(blk, w, 5e-4 if dtype == torch.float32 else 8e-2, l2=dtype != torch.float32)
    finally:
        nn.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dtype", DTYPES)
def test_eva02_reduced_trunk_with_position_embedding_resampling(cuda, dtype):
    """a 2-block member of the EVA02-tiny family (fused qkv with q / v bias, head width 64 -> the online-softmax attention kernels, GluMlp, rotary
    table per call grid): built for 112 x 112 (8 x 8 tokens), called at 98 x 126 (7 x 9 tokens: the position embedding is resampled bilinearly),
    every endpoint and every gradient"""
    from iseg_amd import nn
    from iseg_amd.backbones.eva import Eva

    nn.set_compute_dtype(dtype)
    nn.set_device("cuda:0")
    try:
        eva = Eva(pretrain_img_size=112, pretrain_patch_size=14, patch_size=14, embed_filters=128, depth=2, num_heads=2, qkv_fused=True,
                  mlp_ratio=4 * 2 / 3 * 0.75, swiglu_mlp=True, scale_mlp=False, drop_path_rate=0.2, return_endpoints=True, name="eva_test")
        _setup(eva, torch.empty((2, 112, 112, 3), dtype=torch.float32, device="cuda"))
        f = torch.tensor([0.0, 1.25])
        eva.blocks[1].drop_path_mask = f.cuda()
        g = torch.Generator().manual_seed(0)
        x = torch.randn((2, 98, 126, 3), generator=g)
        ends = eva(x.cuda(), training=True)
        w = {k_: v.requires_grad_(True) for k_, v in OM.export_weights(eva).items()}
        ref = OM.eva_forward(w, x.double(), "eva_test", 2, 2, 14, True, "glu", (8, 8), dp_factors=[None, f.double()])
        assert len(ends) == len(ref) == 4 and tuple(ends[-1].shape) == (2, 7, 9, 128)
        tol = 1e-4 if dtype == torch.float32 else 4e-2
        for i in (1, 2, 3):
            assert _rel(ends[i], ref[i].detach()) < tol, f"endpoint {i}"
        dy = torch.randn(tuple(ref[-1].shape), generator=g)
        ends[-1].backward(dy.cuda().to(dtype))
        ref[-1].backward(dy.to(dtype).double())
        _check_grads(eva, w, 5e-4 if dtype == torch.float32 else 8e-2, l2=dtype != torch.float32)
    finally:
        nn.set_compute_dtype(torch.float32)


def test_eva02_names_are_registered_and_large_builds_with_its_2730_hidden_units(cuda):
    from iseg_amd import nn
    from iseg_amd.backbones.feature_extractor import _builtin_backbones
    from iseg_amd import static_strings as ss

    d = _builtin_backbones()
    assert all(k in d for k in (ss.EVA02_LARGE, ss.EVA02_LARGE_P14, ss.EVA02_TINY, ss.EVA02_LARGE_COCO, ss.EVA02_LARGE_MV))
    nn.set_device("cuda:0")
    tiny = d[ss.EVA02_TINY](return_endpoints=True)
    with nn.dry_run_scope():
        ends = tiny(torch.empty((1, 112, 112, 3), device="cuda"))
    assert len(ends) == 14 and tuple(ends[-1].shape) == (1, 8, 8, 192)
    assert sum(p.numel() for p in tiny.parameters()) == 5464896      # 5.57 M of the published model minus its 24 x 24 position grid (here 8 x 8)
    large = d[ss.EVA02_LARGE](return_endpoints=True)      # 2730 hidden units: the any-width row kernels (test_eva_block_variants covers them)
    with nn.dry_run_scope():
        ends = large(torch.empty((1, 64, 64, 3), device="cuda"))
    assert tuple(ends[-1].shape) == (1, 4, 4, 1024)
    shapes = {p.iseg_name.split("/", 1)[1]: tuple(p.shape) for p in large.parameters() if "/blocks/0/" in p.iseg_name or "blocks_0" in p.iseg_name}
    assert any(sh == (1024, 2730) for sh in shapes.values()) and any(sh == (2730, 1024) for sh in shapes.values()) and any(sh == (2730,) for sh in shapes.values())


def test_eva02_large_forward_backward_is_finite_in_bf16(cuda):
    """the registered large model (24 blocks, 1024 channels, 2730 hidden units, sub-LN over the hidden units) runs forward and backward in bf16 at
    64 x 64 and every parameter receives a finite gradient (its block arithmetic is pinned against the oracle by test_eva_block_variants 3 / 4)"""
    from iseg_amd import nn
    from iseg_amd import static_strings as ss
    from iseg_amd.backbones.feature_extractor import _builtin_backbones
    from iseg_amd.param_store import ParamStore

    nn.set_compute_dtype(torch.bfloat16)
    nn.set_device("cuda:0")
    try:
        large = _builtin_backbones()[ss.EVA02_LARGE](return_endpoints=True)
        with nn.dry_run_scope():
            large(torch.empty((1, 64, 64, 3), device="cuda"))
        store = ParamStore(list(large.parameters()))
        store.sync_shadow()
        g = torch.Generator().manual_seed(0)
        x = torch.randn((1, 64, 64, 3), generator=g).cuda()
        ends = large(x, training=True)
        assert tuple(ends[-1].shape) == (1, 4, 4, 1024) and bool(torch.isfinite(ends[-1].float()).all())
        ends[-1].backward(torch.randn(tuple(ends[-1].shape), generator=g).cuda().to(ends[-1].dtype))
        bad = [p.iseg_name for p in large.parameters() if p.requires_grad and not bool(torch.isfinite(p.grad).all())]
        assert not bad, bad[:5]
        touched = sum(1 for p in large.parameters() if p.requires_grad and float(p.grad.abs().max()) > 0)
        assert touched > 0.9 * sum(1 for p in large.parameters() if p.requires_grad)
    finally:
        nn.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("gate_last,use_norm", [(True, True), (False, False)])
def test_glumlp_with_1x1_conv_projections_equals_the_dense_form(cuda, dtype, gate_last, use_norm):
    """GluMlp(use_conv=True) (backbones/eva/glumlp.py:41-56): fc1 / fc2 as 1 x 1 Conv2D on [N, H, W, C] maps -- the same arithmetic as the Dense form
    on the flattened tokens with kernels reshaped [1, 1, Cin, Cout] <-> [Cin, Cout]: outputs and every gradient agree"""
    from iseg_amd import nn
    from iseg_amd.backbones.eva.mlp import GluMlp

    nn.set_compute_dtype(dtype)
    nn.set_device("cuda:0")
    try:
        N, H, W, C, hid = 2, 6, 5, 32, 64
        nn.set_seed(5)
        conv = GluMlp(hidden_filters=hid, activation="swish", use_conv=True, gate_last=gate_last, use_norm=use_norm, name="glu_conv")
        dense = GluMlp(hidden_filters=hid, activation="swish", use_conv=False, gate_last=gate_last, use_norm=use_norm, name="glu_dense")
        x = rnd((N, H, W, C), 3).to(dtype).cuda()
        xc, xd = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        from iseg_amd.saver.h5_saver import _assign

        _setup(conv, torch.empty((N, H, W, C), dtype=dtype, device="cuda"))
        _setup(dense, torch.empty((N, H * W, C), dtype=dtype, device="cuda"))
        assert tuple(conv.fc1.kernel.shape) == (1, 1, C, hid) and tuple(conv.fc2.kernel.shape) == (1, 1, hid // 2, C)
        pairs = []
        for a, b in ((conv.fc1, dense.fc1), (conv.fc2, dense.fc2)):
            pairs += [(b.kernel, a.kernel.detach().cpu().numpy().reshape(tuple(b.kernel.shape))), (b.bias, a.bias.detach().cpu().numpy())]
        if use_norm:
            pairs += [(dense.norm.gamma, conv.norm.gamma.detach().cpu().numpy()), (dense.norm.beta, conv.norm.beta.detach().cpu().numpy())]
        _assign(pairs)
        yc = conv(xc, training=True)
        yd = dense(xd.reshape(N, H * W, C), training=True)
        assert tuple(yc.shape) == (N, H, W, C)
        tol = 1e-5 if dtype == torch.float32 else 2.0 ** -6
        scale = yd.float().abs().max().item()
        assert (yc.float().reshape(N, H * W, C) - yd.float()).abs().max().item() <= tol * scale
        g = rnd((N, H, W, C), 4).to(dtype).cuda()
        yc.backward(g)
        yd.backward(g.reshape(N, H * W, C))
        assert (xc.grad.float() - xd.grad.float()).abs().max().item() <= tol * xd.grad.float().abs().max().item()
        for a, b in ((conv.fc1, dense.fc1), (conv.fc2, dense.fc2)):
            ga, gb = a.kernel.grad.float().reshape(-1), b.kernel.grad.float().reshape(-1)
            assert (ga - gb).abs().max().item() <= 4 * tol * gb.abs().max().item()
        with pytest.raises(ValueError):
            conv(x.reshape(N, H * W, C))
    finally:
        nn.set_compute_dtype(torch.float32)
