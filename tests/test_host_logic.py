"""Host-side logic that needs no GPU: registry, get_backbone errors, dilation surgery, variable names and the no-weight-decay
rules, schedules, dataset pipeline, flat parameter store, sliding-window tiling."""
import re

import numpy as np

import pytest
import torch


def test_register_backbone_semantics():
    from iseg_amd.backbones import backbone_registry as R

    class MyNet:
        pass

    class Other:
        pass

    R.register_backbone(MyNet)
    assert R.backbone_registry_dict["mynet"] is MyNet
    R.register_backbone(Other, name="mynet")          # first registration wins (backbone_registry.py:17-18)
    assert R.backbone_registry_dict["mynet"] is MyNet
    R.register_backbone(Other, name=("o1", "o2"))
    assert R.backbone_registry_dict["o1"] is Other and R.backbone_registry_dict["o2"] is Other


def test_get_backbone_unknown_raises_value_error():
    from iseg_amd.backbones.feature_extractor import get_backbone

    with pytest.raises(ValueError, match="currently not supported"):
        get_backbone("no_such_net")


def test_registered_backbone_is_constructible_through_get_backbone():
    from iseg_amd.backbones import convnext as cx
    from iseg_amd.backbones.backbone_registry import register_backbone
    from iseg_amd.backbones.feature_extractor import get_backbone

    def convnext_micro(return_endpoints=False):
        return cx.ConvNeXt(depths=[1, 1, 1, 1], filters_list=[16, 32, 64, 128], return_endpoints=return_endpoints)

    register_backbone(convnext_micro, name="convnext_micro_test")
    bb = get_backbone("convnext_micro_test", output_stride=16, return_endpoints=True, image_shape=(1, 64, 64, 3))
    assert bb.downsample_blocks[3].conv.strides == (1, 1) and bb.downsample_blocks[3].conv.dilation_rate == (2, 2)
    assert bb.stages[3].blocks[0].dwconv.dilation_rate == (2, 2)
    assert bb.downsample_blocks[2].conv.strides == (2, 2)
    n = sum(p.numel() for p in bb.parameters())
    assert n > 0


def test_convnext_tiny_aspp_parameter_inventory_and_names():
    from iseg_amd.heads import convnext_tiny_aspp
    from iseg_amd.utils.train_utils import get_no_weight_decay_layers_names_from_model

    m = convnext_tiny_aspp(build_input_size=(64, 64))
    n_backbone = sum(p.numel() for p in m.backbone.parameters())
    assert abs(n_backbone - 27.82e6) < 0.02e6              # 27.82 M
    assert abs(sum(p.numel() for p in m.parameters()) - 33.86e6) < 0.05e6
    names = [p.iseg_name for p in m.parameters()]
    assert "stages/2/8/pwconv1/kernel" in names and "downsample_layers/0/0/kernel" in names and "seg/logits_conv/bias" in names
    assert m.backbone.stages[0].blocks[0].gamma.shape == (96,)
    assert float(m.backbone.stages[0].blocks[0].gamma[0]) == pytest.approx(1e-6)
    excl = get_no_weight_decay_layers_names_from_model(m)
    decayed = [n for n in names if not any(re.search(e, n) for e in excl)]
    assert all(n.endswith("kernel") or n.endswith("/gamma") for n in decayed)
    assert "stages/0/0/pwconv1/kernel" in decayed and "stages/0/0/gamma" in decayed     # layer scale IS decayed (only norm layers excluded)
    assert not any("logits" in n or "/norm/" in n or "/bn/" in n or n.endswith("bias") for n in decayed)
    dp = [b.drop_path_prob for st in m.backbone.stages for b in st.blocks]
    assert dp[0] == 0.0 and dp[-1] == pytest.approx(0.1) and len(dp) == 18


def test_param_store_views_and_segments():
    from iseg_amd.heads import convnext_tiny_aspp
    from iseg_amd.param_store import ParamStore

    m = convnext_tiny_aspp(build_input_size=(64, 64))
    before = {p.iseg_name: p.detach().clone() for p in m.parameters()}
    st = ParamStore(list(m.parameters()))
    assert st.total % 256 == 0 and st.seg_of_block.numel() == st.total // 256
    for p, off, n in st.segments:
        assert off % 256 == 0
        assert p.data.data_ptr() == st.flat_w[off:].data_ptr()
        assert p.grad.data_ptr() == st.flat_g[off:].data_ptr()
        assert torch.equal(p.data, before[p.iseg_name])
        assert torch.equal(p.iseg_compute.float(), p.data.bfloat16().float())
    p0 = st.segments[0][0]
    p0.grad.fill_(3.0)
    assert st.flat_g[:p0.numel()].eq(3.0).all()
    st.zero_grad()
    assert st.flat_g.abs().sum() == 0


def test_schedules_match_oracle():
    from iseg_amd.optimizers.polydecay import WarmUpPolyDecay
    from oracle import tf_ops as O

    d = WarmUpPolyDecay(1e-2, 30000, end_learning_rate=0, warmup_steps=1500, warmup_learning_rate=0)
    for s in (0, 500, 1000, 1500, 2000, 29999, 30000, 40000):
        assert d(s) == pytest.approx(O.warmup_poly_decay(s, 1e-2, 30000, 0.0, 1500, 0.0, 1.0))
    d2 = WarmUpPolyDecay(0.007, 30000, end_learning_rate=0.0, power=0.9)
    assert d2(0) == pytest.approx(0.007) and d2(15000) == pytest.approx(0.007 * 0.5 ** 0.9)


def test_get_optimizer_variants_and_errors():
    from iseg_amd.core_optimizer import get_optimizer
    from iseg_amd.distribution.distribution_utils import Strategy
    from iseg_amd.optimizers import modern

    s = Strategy(one_device=True)
    assert isinstance(get_optimizer(s, optimizer="sgd"), modern.SGD)
    assert isinstance(get_optimizer(s, optimizer="adamw"), modern.AdamW)
    opts = get_optimizer(s, optimizer=["sgd", "adamw"], initial_lr=[0.1, 0.01])
    assert isinstance(opts, list) and len(opts) == 2 and opts[1].learning_rate(0) == pytest.approx(0.01)
    with pytest.raises(ValueError, match="Unsupported optimizer"):
        get_optimizer(s, optimizer="lion")


def test_dataset_pipeline_contract():
    from iseg_amd.data import Dataset, synthetic_batch, synthetic_dataset

    ds = synthetic_dataset(10, 8, 8, seed=0)
    b = next(iter(ds.shuffle(4).repeat().batch(4).prefetch(2)))
    assert b[0].shape == (4, 8, 8, 3) and b[1].shape == (4, 8, 8) and b[1].dtype == torch.int32
    assert len(list(Dataset.from_tensors_list(list(range(10))).batch(4))) == 3
    assert len(list(Dataset.from_tensors_list(list(range(10))).batch(4, drop_remainder=True))) == 2
    assert list(Dataset.from_tensors_list(list(range(6))).shard(2, 1)) == [1, 3, 5]
    x, y = synthetic_batch(2, 32, 32, seed=0)
    assert x.min() >= -1 and x.max() <= 1
    frac = (y == 255).float().mean().item()
    assert 0.05 < frac < 0.15 and int(y[y != 255].max()) <= 20


def test_sliding_window_tiling():
    from iseg_amd.core_inference import get_sliding_start_indexs, get_sliding_window_slices_paddings_list

    assert get_sliding_start_indexs(640, 512) == [0, 128]
    slices, pads, count = get_sliding_window_slices_paddings_list(512, 512, 640, 640)
    assert slices == [[0, 512, 0, 512], [0, 512, 128, 640], [128, 640, 0, 512], [128, 640, 128, 640]]
    assert pads[3] == [128, 0, 128, 0]
    assert int(count.max()) == 4 and int(count.min()) == 1 and int(count[0, 0]) == 1 and int(count[300, 300]) == 4


def test_same_pad_matches_oracle():
    from iseg_amd import kernels as K
    from oracle import tf_ops as O

    for n in (7, 16, 15, 512, 33):
        for k in (1, 2, 3, 4, 7):
            for s in (1, 2, 4):
                for d in (1, 2, 3):
                    out, before, _ = O.same_pad(n, k, s, d)
                    assert K.same_pad(n, k, s, d) == (out, before)


def test_get_scaled_size_pad_mode_1():
    from iseg_amd.utils.common import get_scaled_size

    x = torch.empty(1, 513, 513, 3)
    assert get_scaled_size(x, 0.5, pad_mode=1) == [257, 257]      # int(256.5)=256 even while input odd -> +1
    assert get_scaled_size(x, 1.0, pad_mode=1) == [513, 513]
    assert get_scaled_size(torch.empty(1, 512, 512, 3), 0.75, pad_mode=1) == [384, 384]


# --------------------------------------------------------------------------------------------------------
# Swin / ViT host-side geometry tables (backbones/swin.py, utils/bicubic.py) vs the oracle's tensor-op restatement
# --------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,H,W,ws,shift", [(2, 10, 12, 7, 3), (1, 14, 14, 7, 0), (3, 7, 7, 7, 3), (1, 20, 9, 4, 2)])
def test_swin_window_index_tables_match_pad_roll_partition(N, H, W, ws, shift):
    import torch.nn.functional as TF

    from iseg_amd.backbones.swin import window_index_tables
    from oracle import models as OM

    C = 3
    x = torch.randn(N, H, W, C)
    part, rev, Hp, Wp = window_index_tables(N, H, W, ws, shift)
    flat = x.reshape(-1, C)
    got = torch.where(part.long()[:, None] >= 0, flat[part.long().clamp(min=0)], torch.zeros(1))
    y = TF.pad(x, (0, 0, 0, Wp - W, 0, Hp - H))
    y = torch.roll(y, (-shift, -shift), dims=(1, 2))
    win = OM._window_partition(y, ws).reshape(-1, C)
    assert torch.equal(got, win)
    back = torch.roll(OM._window_reverse(win.reshape(-1, ws, ws, C), ws, Hp, Wp, C), (shift, shift), dims=(1, 2))[:, :H, :W]
    assert torch.equal(win[rev.long()], back.reshape(-1, C))


def test_swin_merge_tables_mask_and_relative_index():
    import torch.nn.functional as TF

    from iseg_amd.backbones.swin import merge_index_tables, relative_position_index, shift_attention_mask
    from oracle import models as OM

    x = torch.randn(2, 5, 7, 3)
    fwd, bwd, H2, W2 = merge_index_tables(2, 5, 7)
    flat = x.reshape(-1, 3)
    got = torch.where(fwd.long()[:, None] >= 0, flat[fwd.long().clamp(min=0)], torch.zeros(1)).reshape(2, H2, W2, 12)
    xp = TF.pad(x, (0, 0, 0, 1, 0, 1))
    ref = torch.cat([xp[:, 0::2, 0::2], xp[:, 1::2, 0::2], xp[:, 0::2, 1::2], xp[:, 1::2, 1::2]], -1)
    assert torch.equal(got, ref)
    assert torch.equal(got.reshape(-1, 3)[bwd.long()], flat)
    for (H, W) in ((14, 14), (19, 23), (7, 7)):
        assert np.array_equal(shift_attention_mask(H, W, 7, 3), OM.swin_attention_mask(H, W, 7, 3).numpy().astype(np.float32))
    idx = relative_position_index((7, 7))
    assert idx.shape == (49, 49) and idx.min() == 0 and idx.max() == 168 and idx[0, 0] == 84 and idx[0, 48] == 0 and idx[48, 0] == 168
    assert np.array_equal(idx, OM._rel_index(7).numpy())


def test_bicubic_matrices_match_oracle_taps():
    from iseg_amd.utils.bicubic import bicubic_matrix
    from oracle import tf_ops as O

    x = torch.randn(1, 24, 24, 3, dtype=torch.float64)
    y = O.resize_bicubic(x, (40, 31))
    Wy, Wx = torch.from_numpy(bicubic_matrix(40, 24)).double(), torch.from_numpy(bicubic_matrix(31, 24)).double()
    y2 = torch.einsum("pw,nowc->nopc", Wx, torch.einsum("oh,nhwc->nowc", Wy, x))
    assert (y - y2).abs().max().item() < 5e-6
    assert np.allclose(bicubic_matrix(24, 24), np.eye(24))
    # 2x up-sampling of a ramp stays a ramp away from the borders (cubic convolution reproduces linear functions)
    r = torch.arange(16, dtype=torch.float64).reshape(1, 16, 1, 1).expand(1, 16, 2, 1)
    up = O.resize_bicubic(r, (32, 2))[0, 4:28, 0, 0]
    assert torch.allclose(up, (torch.arange(4, 28, dtype=torch.float64) + 0.5) / 2 - 0.5, atol=1e-3)


def test_rows2d_views_channel_slices_without_copy():
    """functional._rows2d: the gradient of a channel slice of a wider NHWC tensor reaches the kernels as a strided [rows, C] view"""
    from iseg_amd import functional as F

    big = torch.arange(2 * 3 * 4 * 40, dtype=torch.float32).reshape(2, 3, 4, 40)
    sl = big[..., 8:24]                                    # 16 channels at offset 8: rows 160 B apart, 32-byte aligned start
    v, ld = F._rows2d(sl)
    assert ld == 40 and v.shape == (24, 16) and v.data_ptr() == sl.data_ptr()
    assert torch.equal(v, sl.reshape(24, 16))
    odd = big[..., 3:19]                                   # misaligned start: falls back to a contiguous copy
    v2, ld2 = F._rows2d(odd)
    assert ld2 == 16 and v2.is_contiguous() and torch.equal(v2, odd.reshape(24, 16))
    c, ldc = F._rows2d(big)
    assert ldc == 40 and c.data_ptr() == big.data_ptr()


def test_transposed_kernel_copies_are_a_gpu_bf16_feature():
    """nn.wt() hands out K-contiguous copies only for bf16 GPU shadows; everything else takes the [K][N] path"""
    from iseg_amd import nn

    p = torch.nn.Parameter(torch.randn(8, 16))
    assert nn.wt(p) is None                                # fp32 compute dtype
    nn.set_compute_dtype(torch.bfloat16)
    try:
        assert nn.wt(p) is None                            # no shadow installed
        p.iseg_compute = p.data.to(torch.bfloat16)
        assert nn.wt(p) is None                            # CPU shadow
    finally:
        nn.set_compute_dtype(torch.float32)


def test_convnext_v2_registry_names_shapes_and_surgery():
    """backbones/feature_extractor.py:111-114,148-149 + backbones/convnext_v2.py:236-306: the four V2 names build, GRN variables are
    [1,1,1,4C] zeros named <block>/grn/{gamma,beta}, blocks carry no layer scale, and the dilation surgery edits the same attributes"""
    import numpy as np

    from iseg_amd import static_strings as ss
    from iseg_amd.backbones import convnext_v2 as v2
    from iseg_amd.backbones.feature_extractor import _builtin_backbones, get_backbone

    d = _builtin_backbones()
    assert d[ss.CONVNEXT_V2_NANO] is v2.convnext_v2_nano and d[ss.CONVNEXT_V2_TINY] is v2.convnext_v2_tiny
    assert d[ss.CONVNEXT_V2_LARGE] is v2.convnext_v2_large and d[ss.CONVNEXT_V2_HUGE] is v2.convnext_v2_huge
    bb = get_backbone("convnext_v2_nano", output_stride=8, return_endpoints=True, image_shape=(1, 64, 64, 3))
    shapes = {p.iseg_name: tuple(p.shape) for p in bb.parameters()}
    assert shapes["stages/0/0/grn/gamma"] == (1, 1, 1, 320) and shapes["stages/3/1/grn/beta"] == (1, 1, 1, 2560)
    assert shapes["stages/2/7/pwconv1/kernel"] == (320, 1280) and shapes["stages/2/7/pwconv2/kernel"] == (1280, 320)
    assert "stages/0/0/gamma" not in shapes      # V2 has no layer scale
    assert float(bb.stages[0].blocks[0].grn.gamma.abs().max()) == 0.0
    assert sum(int(np.prod(s)) for s in shapes.values()) == 14981520
    assert [len(s.blocks) for s in bb.stages] == [2, 2, 8, 2]
    rates = [b.drop_path_prob for s in bb.stages for b in s.blocks]
    assert rates == pytest.approx(list(np.linspace(0.0, 0.1, 14)))
    # output_stride 8: stages 2 and 3 run at stride 1 with dilation 2 and 4
    assert bb.downsample_blocks[2].conv.strides == (1, 1) and bb.downsample_blocks[2].conv.dilation_rate == (2, 2)
    assert bb.stages[3].blocks[0].dwconv.dilation_rate == (4, 4) and bb.stages[1].blocks[0].dwconv.dilation_rate == (1, 1)
    huge = v2.convnext_v2_huge()
    assert [len(s.blocks) for s in huge.stages] == [3, 3, 27, 3] and huge.stages[3].blocks[0].filters == 2816


def test_grad_reducer_bucket_layout():
    """the buckets tile the flat gradient buffer exactly once, in model order; the first ones are the small head sizes; a small remainder rides
    the previous bucket (distribution/distribution_utils.py:158-169 is one all-reduce per variable in the reference)"""
    import torch

    from iseg_amd import dist, nn
    from iseg_amd.backbones import convnext as cx
    from iseg_amd.param_store import ParamStore

    nn.set_device("cpu")
    with nn.dry_run_scope():
        net = cx.ConvNeXt(depths=[1, 1, 1, 1], filters_list=[8, 16, 32, 64])
        net(torch.empty(1, 32, 32, 3))
    st = ParamStore(list(net.parameters()))
    total = st.segments[-1][1] + st.padded(st.segments[-1][2])
    for kw in ({"bucket_bytes": 64 << 10, "head_bytes": (4 << 10, 16 << 10)}, {"bucket_bytes": 8 << 10, "head_bytes": ()}, {"bucket_bytes": 1 << 30}):
        red = dist.GradReducer(st, **kw)
        assert red.buckets[0][0] == 0 and red.buckets[-1][1] == total
        for (lo0, hi0, _), (lo1, _, _) in zip(red.buckets, red.buckets[1:]):
            assert hi0 == lo1
        assert sum(c for _, _, c in red.buckets) == len(st.segments)
        for p, off, n in st.segments:
            lo, hi, _ = red.buckets[red.bucket_of[id(p)]]
            assert lo <= off and off + n <= hi
        if len(red.buckets) > 1:      # no tiny trailing collective
            assert (red.buckets[-1][1] - red.buckets[-1][0]) * 4 >= kw["bucket_bytes"] // 4
    red = dist.GradReducer(st, bucket_bytes=64 << 10, head_bytes=(4 << 10, 16 << 10))
    sizes = [(hi - lo) * 4 for lo, hi, _ in red.buckets]
    assert sizes[0] < sizes[-1] and sizes[0] >= 4 << 10


def test_hrnet_registry_and_weight_names():
    """backbones/feature_extractor.py:100-101 + backbones/hrnet.py:541-558: both names build; Keras-style weight names of the nested modules"""
    import numpy as np

    from iseg_amd.backbones.feature_extractor import get_backbone

    bb = get_backbone("hrnet_w48", return_endpoints=True, image_shape=(1, 64, 64, 3))
    shapes = {p.iseg_name: tuple(p.shape) for p in bb.parameters()}
    assert shapes["conv1/kernel"] == (3, 3, 3, 64) and shapes["layer1/0/downsample/0/kernel"] == (1, 1, 64, 256)
    assert shapes["layer1/3/conv3/kernel"] == (1, 1, 64, 256)
    assert shapes["stage2/transition/0/0/kernel"] == (3, 3, 256, 48) and shapes["stage2/transition/1/0/0/kernel"] == (3, 3, 256, 96)
    assert shapes["stage3/transition/2/0/0/kernel"] == (3, 3, 96, 192) and "stage3/transition/0/0/kernel" not in shapes      # same width: passthrough
    assert shapes["stage4/2/fuse_layers/0/3/0/kernel"] == (1, 1, 384, 48)               # 1x1 from branch 3 into branch 0 (then resized up)
    assert shapes["stage4/2/fuse_layers/3/0/0/0/kernel"] == (3, 3, 48, 48)              # three stride-2 steps from branch 0 into branch 3:
    assert shapes["stage4/2/fuse_layers/3/0/2/0/kernel"] == (3, 3, 48, 384)             # ... the last one takes the destination width
    assert shapes["stage4/0/branches/2/1/conv2/kernel"] == (3, 3, 192, 192)
    total = sum(int(np.prod(s)) for s in shapes.values())
    assert 6.4e7 < total < 6.7e7      # HRNet-W48 backbone: ~65 M parameters
    assert len(bb.stages) == 3 and [len(s.modules_list) for s in bb.stages] == [1, 4, 3]


def test_mobilenetv2_names_count_and_atrous_surgery():
    """backbones/mobilenetv2_common.py: 2 223 872 trainable parameters at alpha = 1 (Keras' published count without the top), the Keras weight
    names, and build_atrous_mobilenetv2's stride / rate edits (:204-222)"""
    import numpy as np

    from iseg_amd.backbones.feature_extractor import get_backbone
    from iseg_amd.backbones.mobilenetv2_common import _make_divisible, correct_pad

    bb = get_backbone("mobilenetv2", output_stride=8, return_endpoints=True, image_shape=(1, 64, 64, 3))
    shapes = {p.iseg_name: tuple(p.shape) for p in bb.parameters()}
    assert sum(int(np.prod(s)) for s in shapes.values()) == 2223872
    assert shapes["Conv1/kernel"] == (3, 3, 3, 32) and shapes["expanded_conv_depthwise/depthwise_kernel"] == (3, 3, 32, 1)
    assert "expanded_conv_expand/kernel" not in shapes and shapes["block_1_expand/kernel"] == (1, 1, 16, 96)
    assert shapes["block_16_project/kernel"] == (1, 1, 960, 320) and shapes["Conv_1/kernel"] == (1, 1, 320, 1280)
    sr = [(b.strides, b.atrous_rates) for b in bb.blocks]
    assert sr[1] == (2, 1) and sr[3] == (2, 1) and sr[6] == (1, 2) and sr[7] == (1, 2) and sr[13] == (1, 4) and sr[16] == (1, 4)
    assert _make_divisible(32 * 0.35, 8) == 16 and _make_divisible(24 * 1.4, 8) == 32
    assert correct_pad((None, 64, 63, 8), 3) == ((0, 1), (1, 1))


def test_dcnv3_joint_form_is_refused_for_strided_layers():
    """round-5 advisor: the joint offset | mask projection has one row per INPUT pixel of x1, the sampling kernel wants one per OUTPUT pixel: a
    stride > 1 layer (or any geometry whose output map differs from x1's) must take the layer-by-layer route, not raise from the kernel wrapper"""
    import types

    import torch

    from iseg_amd import functional as F

    lay = types.SimpleNamespace(kernel=torch.zeros(64, 54), bias=torch.zeros(54))
    x1 = torch.zeros(1, 16, 16, 64, dtype=torch.bfloat16)
    assert F.dcnv3_joint_ok(x1, lay, lay, (3, 3), 1, (16, 16)) == bool(F._DCN_JOINT)
    assert not F.dcnv3_joint_ok(x1, lay, lay, (3, 3), 2, (8, 8))
    assert not F.dcnv3_joint_ok(x1, lay, lay, (3, 3), 1, (14, 14))      # pad 0: the output map shrinks


def test_oracle_thread_budget_follows_the_cgroup_quota(monkeypatch):
    """oracle/host_threads.py (test infrastructure): the CPU oracle's thread count is the affinity mask capped by the cgroup CPU quota -- a GPU box shows
    256 logical cores under a quota of 16, and torch's own choice (128 threads) ran the fp64 oracle 8 x slower"""
    import builtins
    import io
    import os

    from oracle import host_threads

    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            return io.StringIO(fake["cpu.max"])
        return real_open(path, *a, **k)

    fake = {"cpu.max": "1600000 100000\n"}
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)))
    monkeypatch.setattr(builtins, "open", fake_open)
    assert host_threads.cpu_budget() == 16
    fake["cpu.max"] = "max 100000\n"
    assert host_threads.cpu_budget() == 256
    fake["cpu.max"] = "150000 100000\n"      # a quota of one and a half cores
    assert host_threads.cpu_budget() == 2
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(4)))
    fake["cpu.max"] = "1600000 100000\n"
    assert host_threads.cpu_budget() == 4
