"""HIP kernels against the committed golden vectors (tests/golden/oracle_ops.npz): fixed inputs, fixed expected outputs, no
oracle code in the loop."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_ops.npz"))


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).cuda()


def near(got, want, tol):
    err = np.abs(got.detach().cpu().double().numpy() - np.asarray(want, dtype=np.float64)).max()
    assert err <= tol * max(1.0, np.abs(want).max()), err


@pytest.fixture(autouse=True)
def _fp32():
    from iseg_amd import nn

    nn.set_compute_dtype(torch.float32)
    yield
    nn.set_compute_dtype(torch.float32)


def test_conv_depthwise_norm_activation(cuda):
    from iseg_amd import functional as F
    from iseg_amd import kernels as K

    x, k, b = dev(G["conv_x"]), torch.nn.Parameter(dev(G["conv_k"])), torch.nn.Parameter(dev(G["conv_b"]))
    near(F.conv2d(x, k, b, (1, 1), (2, 2), "same"), G["conv_s1d2"], 2e-5)
    near(F.conv2d(x, k, b, (2, 2), (1, 1), "same"), G["conv_s2d1"], 2e-5)
    yd = K.dwconv2d(dev(G["dw_x"]), dev(G["dw_k"]).reshape(49, 16), None, 7, 1, 3, 3)
    near(yd, G["dw_y"], 2e-5)
    y, _, _ = K.layernorm_fwd(dev(G["dw_y"]).reshape(-1, 16), dev(G["ln_gamma"]), dev(G["ln_beta"]), 1e-6)
    near(y.reshape(G["ln_y"].shape), G["ln_y"], 2e-5)
    near(K.act_fwd(dev(G["dw_y"]), K.ACT_GELU), G["gelu_y"], 2e-6)


def test_resize_pool_norm_variants(cuda):
    from iseg_amd import functional as F
    from iseg_amd import kernels as K

    near(K.resize_bilinear(dev(G["resize_x"]), 13, 9), G["resize_y"], 2e-6)
    assert np.array_equal(K.resize_nearest_i32(dev(G["nearest_x"], torch.int32), 4, 9).cpu().numpy(), G["nearest_y"])
    y, _, _ = K.groupnorm_fwd(dev(G["gn_x"]).reshape(2, 9, 12), None, None, 3, 1e-3)
    near(y.reshape(G["gn_y"].shape), G["gn_y"], 2e-5)
    y, _ = K.rmsnorm_fwd(dev(G["gn_x"]).reshape(-1, 12), torch.zeros(12, device="cuda"), 1e-6)
    near(y.reshape(G["rms_y"].shape), G["rms_y"], 2e-5)
    near(F.max_pool2d(dev(G["pool_x"]), 3, 2, "same"), G["maxpool_3s2"], 1e-7)
    near(F.avg_pool2d(dev(G["pool_x"]), 2, 2, "same"), G["avgpool_2s2"], 1e-6)


def test_loss_argmax_confusion(cuda):
    from iseg_amd import kernels as K

    logits = dev(G["ce_logits"]).reshape(-1, 21)
    labels = dev(G["ce_labels"], torch.int32).reshape(-1)
    px, _, _ = K.softmax_ce_ignore(logits, labels, 255, want_px=True)
    near(px, G["ce_px"].reshape(-1), 2e-5)
    cm = torch.zeros(21 * 21, dtype=torch.int64, device="cuda")          # uint64 counts, exact
    pred = K.argmax_confusion(logits, labels, 255, cm=cm, want_pred=True)
    assert np.array_equal(pred.cpu().numpy().reshape(G["argmax"].shape), G["argmax"])          # bit-exact index work
    assert np.array_equal(cm.cpu().numpy().reshape(21, 21), G["confusion"].astype(np.int64))


def test_dcnv3_and_bicubic(cuda):
    from iseg_amd import functional as F
    from iseg_amd import kernels as K
    from iseg_amd.utils.bicubic import bicubic_matrix

    y = K.dcnv3_fwd(dev(G["dcn_x"]), dev(G["dcn_off"]), dev(G["dcn_mask"]), 2, 4, 3, 3, 1, 1, 1, 1.0)
    near(y, G["dcn_y"], 2e-5)
    pos = torch.nn.Parameter(torch.cat([torch.zeros(1, 1, 8), torch.from_numpy(G["bicubic_x"]).float().reshape(1, 16, 8)], 1).cuda())
    wy, wx = torch.from_numpy(bicubic_matrix(6, 4)).cuda(), torch.from_numpy(bicubic_matrix(5, 4)).cuda()
    out = F.resize_pos_embed(pos, wy, wx, 1, torch.float32)
    near(out[0, 1:].reshape(6, 5, 8), G["bicubic_y"][0], 2e-5)


def test_host_tables_and_schedules():
    from iseg_amd.backbones.swin import relative_position_index, shift_attention_mask
    from iseg_amd.core_inference import get_sliding_start_indexs
    from iseg_amd.optimizers.polydecay import WarmUpPolyDecay

    assert np.array_equal(shift_attention_mask(19, 23, 7, 3), G["swin_mask_19x23"])
    assert np.array_equal(relative_position_index((7, 7)), G["swin_rel_index"])
    assert list(get_sliding_start_indexs(640, 512)) == list(G["sliding_640_512"])
    assert list(get_sliding_start_indexs(1024, 512)) == list(G["sliding_1024_512"])
    d = WarmUpPolyDecay(1e-2, 30000, end_learning_rate=0.0, warmup_steps=1500, warmup_learning_rate=0.0, power=1.0)
    got = [float(d(s)) for s in (0, 500, 1000, 1500, 2000, 29999)]
    assert np.allclose(got, G["poly_lr"], rtol=1e-6, atol=1e-12)
