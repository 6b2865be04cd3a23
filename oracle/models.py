"""CPU restatement of the model compositions on top of oracle/tf_ops.py.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see
oracle/tf_ops.py).  Follows backbones/convnext.py:47-63,82-91,176-191, layers/model_builder.py:79-98,260-273,
layers/aspp.py:57-71, layers/core_model_ext.py:185-256 of the reference.  Weights are addressed by the same slash names the
product model uses (Keras layouts), so a weight dict exported from iseg_amd drives this code directly."""
import torch

from . import tf_ops as O


def export_weights(model, dtype=torch.float64):
    out = {}
    for p in model.parameters():
        out[p.iseg_name] = p.detach().cpu().to(dtype).clone()
    for b in model.buffers():
        n = getattr(b, "iseg_name", None)
        if n is not None:
            out[n] = b.detach().cpu().to(dtype).clone()
    return out


def conv_norm_act(w, prefix, x, training, dilation=1, relu=True, bn_eps=1e-3, bn_momentum=0.9, new_stats=None):
    """ConvNormAct: conv(same, no bias) -> BN -> ReLU   (dropout is identity in parity runs)"""
    y = O.conv2d(x, w[f"{prefix}/conv/kernel"], w.get(f"{prefix}/conv/bias"), 1, dilation, "same")
    g, b = w[f"{prefix}/bn/gamma"], w[f"{prefix}/bn/beta"]
    if training:
        y, mean, var = O.batch_norm_train(y, g, b, bn_eps)
        if new_stats is not None:
            new_stats[f"{prefix}/bn/moving_mean"] = O.moving_update(w[f"{prefix}/bn/moving_mean"], mean.detach(), bn_momentum)
            new_stats[f"{prefix}/bn/moving_variance"] = O.moving_update(w[f"{prefix}/bn/moving_variance"], var.detach(), bn_momentum)
    else:
        y = O.batch_norm_infer(y, g, b, w[f"{prefix}/bn/moving_mean"], w[f"{prefix}/bn/moving_variance"], bn_eps)
    return torch.relu(y) if relu else y


def convnext_block(w, prefix, x, dilation=1, dp_factor=None):
    y = O.depthwise_conv2d(x, w[f"{prefix}/dwconv/depthwise_kernel"], w[f"{prefix}/dwconv/bias"], 1, dilation)
    y = O.layer_norm(y, w[f"{prefix}/norm/gamma"], w[f"{prefix}/norm/beta"], 1e-6)
    y = O.gelu(O.dense(y, w[f"{prefix}/pwconv1/kernel"], w[f"{prefix}/pwconv1/bias"]))
    y = O.dense(y, w[f"{prefix}/pwconv2/kernel"], w[f"{prefix}/pwconv2/bias"])
    if f"{prefix}/gamma" in w:
        y = y * w[f"{prefix}/gamma"]
    if dp_factor is not None:
        y = O.drop_path(y, dp_factor)
    return y + x


def convnext_backbone(w, x, depths=(3, 3, 9, 3), output_stride=32, dp_factors=None):
    """returns [None, s0, s1, s2, s3]; dilation surgery as build_dilated_convnext (backbones/convnext.py:245-266)"""
    endpoints = [None]
    current_os, current_dil = 1, 1
    blk = 0
    for i, depth in enumerate(depths):
        stride = 4 if i == 0 else 2
        dil_conv, dil_dw, s = 1, 1, stride
        if current_os >= output_stride:
            current_dil *= stride
            s, dil_conv, dil_dw = 1, current_dil, current_dil
        else:
            current_os *= stride
        pc, pn = (f"downsample_layers/{i}/0", f"downsample_layers/{i}/1") if i == 0 else (f"downsample_layers/{i}/1", f"downsample_layers/{i}/0")
        if i == 0:
            x = O.conv2d(x, w[f"{pc}/kernel"], w[f"{pc}/bias"], s, dil_conv, "same")
            x = O.layer_norm(x, w[f"{pn}/gamma"], w[f"{pn}/beta"], 1e-6)
        else:
            x = O.layer_norm(x, w[f"{pn}/gamma"], w[f"{pn}/beta"], 1e-6)
            x = O.conv2d(x, w[f"{pc}/kernel"], w[f"{pc}/bias"], s, dil_conv, "same")
        for j in range(depth):
            x = convnext_block(w, f"stages/{i}/{j}", x, dil_dw, None if dp_factors is None else dp_factors[blk])
            blk += 1
        endpoints.append(x)
    return endpoints


def aspp(w, prefix, x, training, rates=(3, 6, 9), new_stats=None):
    N, H, W, C = x.shape
    img = x.mean(dim=(1, 2), keepdim=True)
    img = conv_norm_act(w, f"{prefix}/image_level_block/conv", img, training, new_stats=new_stats)
    img = img.expand(N, H, W, img.shape[-1])
    outs = [img, conv_norm_act(w, f"{prefix}/pixel_level_block", x, training, new_stats=new_stats)]
    for r in rates:
        outs.append(conv_norm_act(w, f"{prefix}/asp_convs_{r}", x, training, dilation=r, new_stats=new_stats))
    return torch.cat(outs, dim=-1)


def convnext_aspp_forward(w, x, training=False, output_stride=32, dp_factors=None, depths=(3, 3, 9, 3), head="aspp_head", seg="seg",
                          new_stats=None):
    """SegManaged._call_internal with the ASPP head composition (iseg_amd/heads.py)."""
    ends = convnext_backbone(w, x, depths, output_stride, dp_factors)
    mult = max(32 // output_stride, 1)
    feat = aspp(w, f"{head}/aspp", ends[-1], training, rates=tuple(r * mult for r in (3, 6, 9)), new_stats=new_stats)
    feat = conv_norm_act(w, f"{head}/end_conv", feat, training, new_stats=new_stats)
    small = O.conv2d(feat, w[f"{seg}/logits_conv/kernel"], w[f"{seg}/logits_conv/bias"], 1, 1, "same")
    logits = O.resize_bilinear(small, (x.shape[1], x.shape[2]))
    return {"endpoints": ends, "head": feat, "small_logits": small, "logits": logits}


def mean_ce_loss(logits, labels, num_class=21, ignore_label=255, class_weights=None):
    """Keras: mean over ALL positions of the NONE-reduced weighted loss"""
    return O.softmax_ce_ignore(labels, logits, num_class, ignore_label, class_weights).mean()


# ------------------------------------------------------------------------------------------------------
# layers/fpn.py:16-61 FeaturePyramidNetwork, layers/simpledecoder.py:8-36 SimpleDecoder
# ------------------------------------------------------------------------------------------------------
def fpn(w, prefix, feats, training, new_stats=None):
    x = feats[-1]
    outs = [x]
    for i in range(len(feats) - 2, -1, -1):
        skip = conv_norm_act(w, f"{prefix}/skip_conv_filters{i}", O.replace_nan_or_inf(feats[i], 0.0), training, new_stats=new_stats)
        x = O.resize_bilinear(x, skip.shape[1:3]) + skip
        outs.append(x)
    outs.reverse()
    return outs


def simple_decoder(w, prefix, low, high, training, new_stats=None):
    low = conv_norm_act(w, f"{prefix}/low_level_entry_conv", low, training, new_stats=new_stats)
    x = torch.cat([low, O.resize_bilinear(high, low.shape[1:3])], dim=-1)
    x = conv_norm_act(w, f"{prefix}/finetune_conv0", x, training, new_stats=new_stats)
    return conv_norm_act(w, f"{prefix}/finetune_conv1", x, training, new_stats=new_stats)


# ------------------------------------------------------------------------------------------------------
# ResNet "slim/beta" as get_backbone builds it (feature_extractor.py:58-66,139-141; resnet_common.py:94-184,245-345,
# 523-598; resnet_blocks.py:111-205): 3x3 deep stem, max-pool 3x3/s2 SAME, Stack2 (stride in the LAST block of stacks
# 0..2), BlockType2 bottleneck with average-pooled identity shortcut, atrous surgery + multi-grid on the last stack.
# ------------------------------------------------------------------------------------------------------
def _bn(w, prefix, y, training, eps, momentum=0.9, new_stats=None):
    g, b = w[f"{prefix}/gamma"], w[f"{prefix}/beta"]
    if training:
        y, mean, var = O.batch_norm_train(y, g, b, eps)
        if new_stats is not None:
            new_stats[f"{prefix}/moving_mean"] = O.moving_update(w[f"{prefix}/moving_mean"], mean.detach(), momentum)
            new_stats[f"{prefix}/moving_variance"] = O.moving_update(w[f"{prefix}/moving_variance"], var.detach(), momentum)
        return y
    return O.batch_norm_infer(y, g, b, w[f"{prefix}/moving_mean"], w[f"{prefix}/moving_variance"], eps)


def resnet_plan(num_of_blocks=(3, 4, 6, 3), output_stride=32, multi_grids=(1, 2, 4)):
    """per stack, per block: (stride, dilation, conv_shortcut) after build_atrous_resnet + apply_multi_grid(block_index=-1)"""
    plan = []
    for si, nb in enumerate(num_of_blocks):
        stride1 = [2, 2, 2, 1][si]
        blocks = []
        for bi in range(nb):
            last = bi == nb - 1
            blocks.append([stride1 if last else 1, 1, bi == 0])
        plan.append(blocks)
    current_os, rate = 4, 1
    for blocks in plan:
        for blk in blocks:
            if blk[0] > 1:
                if current_os >= output_stride:
                    rate *= 2
                    blk[0] = 1
                    blk[1] = blk[1] * rate
                else:
                    current_os *= 2
            else:
                blk[1] = blk[1] * rate
    for bi, blk in enumerate(plan[-1]):
        blk[1] = blk[1] * multi_grids[bi]
    return plan


def resnet_block2(w, name, x, stride, dilation, conv_shortcut, training, eps=1.001e-5, new_stats=None):
    shortcut = x
    if conv_shortcut:
        shortcut = _bn(w, f"{name}_0_bn", O.conv2d(x, w[f"{name}_0_conv/kernel"], None, stride, 1, "valid"), training, eps,
                       new_stats=new_stats)
    if stride > 1:
        shortcut = O.avg_pool_same(shortcut, stride, stride)
    y = torch.relu(_bn(w, f"{name}_1_bn", O.conv2d(x, w[f"{name}_1_conv/kernel"], None, 1, 1, "valid"), training, eps, new_stats=new_stats))
    y = torch.relu(_bn(w, f"{name}_2_bn", O.conv2d(y, w[f"{name}_2_conv/kernel"], None, stride, dilation, "same"), training, eps,
                       new_stats=new_stats))
    y = _bn(w, f"{name}_3_bn", O.conv2d(y, w[f"{name}_3_conv/kernel"], None, 1, 1, "valid"), training, eps, new_stats=new_stats)
    return torch.relu(shortcut + y)


def resnet_forward(w, x, num_of_blocks=(3, 4, 6, 3), output_stride=32, multi_grids=(1, 2, 4), training=False, new_stats=None):
    eps = 1.001e-5
    for i, s in ((1, 2), (2, 1), (3, 1)):
        x = torch.relu(_bn(w, f"conv1_{i}_bn", O.conv2d(x, w[f"conv1_{i}_conv/kernel"], None, s, 1, "same"), training, eps,
                           new_stats=new_stats))
    endpoints = [x]
    x = O.max_pool_same(x, 3, 1 if output_stride == 2 else 2)
    plan = resnet_plan(num_of_blocks, output_stride, multi_grids)
    for si, blocks in enumerate(plan):
        emits = [2, 2, 2, 1][si] > 1          # Stack2.output_endpoint is fixed at construction (stride1 > 1)
        for bi, (stride, dil, conv_sc) in enumerate(blocks):
            if bi == len(blocks) - 1 and emits:
                endpoints.append(x)           # value before the (possibly removed) stride
            x = resnet_block2(w, f"conv{si + 2}_block{bi + 1}", x, stride, dil, conv_sc, training, eps, new_stats)
    endpoints.append(x)
    return endpoints
