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


def convnext_backbone(w, x, depths=(3, 3, 9, 3), output_stride=32, dp_factors=None, block=None):
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
            x = (block or convnext_block)(w, f"stages/{i}/{j}", x, dil_dw, None if dp_factors is None else dp_factors[blk])
            blk += 1
        endpoints.append(x)
    return endpoints


def convnext_v2_block(w, prefix, x, dilation=1, dp_factor=None):
    """backbones/convnext_v2.py:83-98 Block.call: no layer scale, GRN between the GELU and the second pointwise product"""
    y = O.depthwise_conv2d(x, w[f"{prefix}/dwconv/depthwise_kernel"], w[f"{prefix}/dwconv/bias"], 1, dilation)
    y = O.layer_norm(y, w[f"{prefix}/norm/gamma"], w[f"{prefix}/norm/beta"], 1e-6)
    y = O.gelu(O.dense(y, w[f"{prefix}/pwconv1/kernel"], w[f"{prefix}/pwconv1/bias"]))
    y = O.grn(y, w[f"{prefix}/grn/gamma"], w[f"{prefix}/grn/beta"], 1e-6)
    y = O.dense(y, w[f"{prefix}/pwconv2/kernel"], w[f"{prefix}/pwconv2/bias"])
    if dp_factor is not None:
        y = O.drop_path(y, dp_factor)
    return y + x


def convnext_v2_backbone(w, x, depths=(2, 2, 8, 2), output_stride=32, dp_factors=None):
    """backbones/convnext_v2.py:205-218 with the dilation surgery of :284-306 (same walk as convnext_backbone)"""
    return convnext_backbone(w, x, depths, output_stride, dp_factors, block=convnext_v2_block)


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


# ------------------------------------------------------------------------------------------------------
# layers/multihead_self_attention.py:170-203
# ------------------------------------------------------------------------------------------------------
def mhsa_layer(w, prefix, x, heads):
    def conv1x1(name, t):
        return O.conv2d(t, w[f"{prefix}/{name}/kernel"], w.get(f"{prefix}/{name}/bias"), 1, 1, "valid")

    q, k, v = conv1x1("query_conv", x), conv1x1("key_conv", x), conv1x1("value_conv", x)
    return O.mhsa_core(q, k, v, heads)


def nasfpn_forward(w, inputs, name, block_specs, min_level=3, max_level=7, num_filters=256, num_repeats=5, use_sum_for_combination=True,
                   training=False, eps=1e-3, use_separable_conv=False, activation="relu"):
    """layers/nasfpn.py:196-232 (input pyramid), :248-271 (resample), :304-311 (global attention), :313-383 (one cell), line by line.
    inputs: {level: [N, H, W, C]}; block_specs: [(level, combine_fn, (offset0, offset1), is_output)].  use_separable_conv (:176-181): every
    convolution is keras SeparableConv2D = depthwise k x k (no bias) then pointwise 1 x 1 + bias; activation (:194): keras.activations.get."""
    act = {"relu": torch.relu, "swish": lambda t: t * torch.sigmoid(t), "silu": lambda t: t * torch.sigmoid(t), "gelu": O.gelu}[activation]

    def conv(prefix, t):
        if use_separable_conv:
            t = O.depthwise_conv2d(t, w[f"{prefix}/depthwise_kernel"], None, 1, 1, "same")
            return O.conv2d(t, w[f"{prefix}/pointwise_kernel"], w[f"{prefix}/bias"], 1, 1, "same")
        return O.conv2d(t, w[f"{prefix}/kernel"], w[f"{prefix}/bias"], 1, 1, "same")

    def bn(prefix, y):
        g, b = w[f"{prefix}/gamma"], w[f"{prefix}/beta"]
        if training:
            return O.batch_norm_train(y, g, b, eps)[0]
        return O.batch_norm_infer(y, g, b, w[f"{prefix}/moving_mean"], w[f"{prefix}/moving_variance"], eps)

    def resample(x, input_level, target_level):
        if input_level < target_level:
            stride = 2 ** (target_level - input_level)
            return O.max_pool_same(x, stride, stride)
        if input_level > target_level:
            s = 2 ** (input_level - target_level)
            return x.repeat_interleave(s, dim=1).repeat_interleave(s, dim=2)      # nearest_upsampling (:48-84)
        return x

    def global_attention(feat0, feat1):
        m = torch.sigmoid(feat0.amax(dim=(1, 2), keepdim=True))
        return feat0 + feat1 * m

    feats = []
    for level in range(min_level, max_level + 1):
        if level in inputs:
            x = inputs[level]
            if x.shape[-1] != num_filters:
                p = f"{name}/resample_l{level}"
                x = bn(f"{p}/bn", conv(f"{p}/separable_conv2d", x))
            feats.append(x)
        else:
            feats.append(O.max_pool_same(feats[-1], 2, 2))
    n_levels = max_level - min_level + 1
    out = None
    for r in range(num_repeats):
        feats = list(feats)
        levels = list(range(min_level, max_level + 1))
        used = [0] * len(feats)
        for i, (new_level, combine_fn, (i0, i1), is_output) in enumerate(block_specs):
            node0, l0, node1, l1 = feats[i0], levels[i0], feats[i1], levels[i1]
            used[i0] += 1
            used[i1] += 1
            node0, node1 = resample(node0, l0, new_level), resample(node1, l1, new_level)
            if use_sum_for_combination or combine_fn == "sum":
                new_node = node0 + node1
            else:
                new_node = global_attention(node0, node1) if l0 >= l1 else global_attention(node1, node0)
            if is_output:
                for j in range(len(feats)):
                    if used[j] == 0 and levels[j] == new_level:
                        used[j] += 1
                        new_node = new_node + feats[j]
            p = f"{name}/cell_{r}/sub_policy{i}/op_after_combine{n_levels + i}"
            new_node = bn(f"{p}/bn", conv(f"{p}/conv", act(new_node)))
            feats.append(new_node)
            levels.append(new_level)
            used.append(0)
        out = {levels[i]: feats[i] for i in range(len(feats) - n_levels, len(feats))}
        feats = [out[level] for level in range(min_level, max_level + 1)]
    return out


# ------------------------------------------------------------------------------------------------------
# backbones/moat/*: stem (moat.py:113-137), MBConvBlock / MOATBlock (moat_blocks.py:216-245, 466-508), Attention without relative position embedding
# (attention.py:318-339), MOAT.call (moat.py:227-242)
# ------------------------------------------------------------------------------------------------------
def _moat_mbconv(w, p, x, stride, with_se, training, new_stats, bn_eps=1e-3):
    shortcut = O.avg_pool_same(x, 2, stride) if stride > 1 else x
    if f"{p}/shortcut_conv/kernel" in w:
        shortcut = O.conv2d(shortcut, w[f"{p}/shortcut_conv/kernel"], w[f"{p}/shortcut_conv/bias"], 1, 1, "same")
    y = _bn(w, f"{p}/pre_norm", x, training, bn_eps, new_stats=new_stats)
    y = O.conv2d(y, w[f"{p}/expand_conv/kernel"], None, 1, 1, "same")
    y = O.gelu(_bn(w, f"{p}/expand_norm", y, training, bn_eps, new_stats=new_stats))
    y = O.depthwise_conv2d(y, w[f"{p}/depthwise_conv/depthwise_kernel"], None, stride, 1, "same")
    y = O.gelu(_bn(w, f"{p}/depthwise_norm", y, training, bn_eps, new_stats=new_stats))
    if with_se:
        g = y.mean(dim=(1, 2), keepdim=True)
        g = O.conv2d(g, w[f"{p}/se/reduce_conv2d/kernel"], w[f"{p}/se/reduce_conv2d/bias"], 1, 1, "same")
        g = O.conv2d(g * torch.sigmoid(g), w[f"{p}/se/expand_conv2d/kernel"], w[f"{p}/se/expand_conv2d/bias"], 1, 1, "same")
        y = torch.sigmoid(g) * y
    return O.conv2d(y, w[f"{p}/shrink_conv/kernel"], w[f"{p}/shrink_conv/bias"], 1, 1, "same"), shortcut


def moat_position_bias(table, height, width, resize):
    """backbones/moat/attention.py:68-120,258-306: the [heads, e_h, e_w] table, resized bilinearly to [2 h - 1, 2 w - 1] when the layer has a scale
    ratio, then re-indexed by two one-hot lookups (max relative distance h - 1 / w - 1): bias[n, (i, j), (x, y)] = R[n, x - i + h - 1, y - j + w - 1].
    The reference evaluates this once, in build() (a constant: detached here)."""
    r = table.detach()
    if resize:
        r = O.resize_bilinear(r.unsqueeze(-1), (2 * height - 1, 2 * width - 1)).squeeze(-1)
    hl = torch.zeros(height, height, 2 * height - 1, dtype=r.dtype)
    for i in range(height):
        for x in range(height):
            hl[i, x, x - i + height - 1] = 1
    wl = torch.zeros(width, width, 2 * width - 1, dtype=r.dtype)
    for j in range(width):
        for y in range(width):
            wl[j, y, y - j + width - 1] = 1
    t = torch.einsum("nhw,ixh->nixw", r, hl)
    t = torch.einsum("nixw,jyw->nijxy", t, wl)
    return t.reshape(r.shape[0], height * width, height * width)


def _moat_attention(w, p, x, head_size):
    B, H, W, C = x.shape
    t = x.reshape(B, H * W, C)
    q = torch.einsum("btc,cnk->btnk", t, w[f"{p}/q/weight"]) + w[f"{p}/q/bias"]
    k = torch.einsum("btc,cnk->btnk", t, w[f"{p}/k/weight"]) + w[f"{p}/k/bias"]
    v = torch.einsum("btc,cnk->btnk", t, w[f"{p}/v/weight"]) + w[f"{p}/v/bias"]
    logits = torch.einsum("bsnk,btnk->bnst", q * head_size ** -0.5, k)
    table = w.get(f"{p}/relative_position_embedding")
    if table is not None:      # (:320-323) the table is resized whenever the block hands a scale ratio over, i.e. whenever it exists in a MOAT block
        logits = logits + moat_position_bias(table, H, W, resize=tuple(table.shape[1:]) != (2 * H - 1, 2 * W - 1))
    a = torch.softmax(logits, dim=-1)
    o = torch.einsum("bnst,btnk->bsnk", a, v)
    return (torch.einsum("bsnk,nkc->bsc", o, w[f"{p}/o/weight"]) + w[f"{p}/o/bias"]).reshape(B, H, W, -1)


def _moat_windowed_attention(w, p, x, head_size, window):
    """backbones/moat/moat_blocks.py:407-434,486-497: _make_windows -> attention inside each window -> _remove_windows"""
    B, H, W, C = x.shape
    wh, ww = window
    t = x.reshape(B, H // wh, wh, W // ww, ww, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, wh, ww, C)
    a = _moat_attention(w, p, t, head_size)
    a = a.reshape(B, H // wh, W // ww, wh, ww, -1).permute(0, 1, 3, 2, 4, 5)
    return a.reshape(B, H, W, -1)


def moat_forward(w, x, name, block_types, num_blocks, stage_stride=(2, 2, 2, 2), head_size=32, stem=2, training=False, dp_factors=None,
                 new_stats=None, ln_eps=1e-5, window_size=None):
    """MOAT.call with return_endpoints=True: [stem, stage 0 .. 3]; dp_factors[block name] = per-sample factors (one vector for an MBConv block, a
    pair for a MOAT block) or absent; window_size[stage] = [height, width] or None (MOAT stages only, moat.py:177-209)"""
    dp_factors = dp_factors or {}
    window_size = window_size or [None] * len(block_types)
    for i in range(stem):
        x = O.conv2d(x, w[f"{name}/stem/conv_{i}/kernel"], w[f"{name}/stem/conv_{i}/bias"], 2 if i == 0 else 1, 1, "same")
        if i < stem - 1:
            x = O.gelu(_bn(w, f"{name}/stem/norm_{i}", x, training, 1e-3, new_stats=new_stats))
    ends = [x]
    for s, kind in enumerate(block_types):
        for b in range(num_blocks[s]):
            p = f"{name}/block_{s:0>2d}_{b:0>2d}"
            stride = stage_stride[s] if b == 0 else 1
            y, shortcut = _moat_mbconv(w, p, x, stride, kind == "mbconv", training, new_stats)
            f = dp_factors.get(p)
            if kind == "mbconv":
                x = shortcut + (y if f is None else y * f.reshape(-1, 1, 1, 1))
            else:
                x = shortcut + (y if f is None else y * f[0].reshape(-1, 1, 1, 1))
                normed = O.layer_norm(x, w[f"{p}/attention_norm/gamma"], w[f"{p}/attention_norm/beta"], ln_eps)
                if window_size[s]:
                    a = _moat_windowed_attention(w, f"{p}/attention", normed, head_size, window_size[s])
                else:
                    a = _moat_attention(w, f"{p}/attention", normed, head_size)
                x = x + (a if f is None else a * f[1].reshape(-1, 1, 1, 1))
        ends.append(x)
    return ends


def se_module(w, p, x, bias=True, activation=torch.relu):
    """layers/se.py:33-47"""
    g = x.mean(dim=(1, 2), keepdim=True)
    g = activation(O.conv2d(g, w[f"{p}/down_conv/kernel"], w.get(f"{p}/down_conv/bias") if bias else None, 1, 1, "valid"))
    g = torch.sigmoid(O.conv2d(g, w[f"{p}/expand_conv/kernel"], w.get(f"{p}/expand_conv/bias") if bias else None, 1, 1, "valid"))
    return g * x


def fapn_forward(w, name, feats):
    """layers/fapn.py:13-140: FeatureAlignedPyramidNet(warp_coarse_feature=False) on feats (fine -> coarse); returns all levels fine -> coarse"""
    x = feats[-1]
    out = [x]
    for i in range(len(feats) - 2, -1, -1):
        p = f"{name}/skip_conv_filters{i}"
        large = feats[i]
        up = O.resize_bilinear(x, (large.shape[1], large.shape[2]))
        arm = se_module(w, f"{p}/lateral_conv", large, bias=False) + large                  # FeatureSelectionModule :31-41
        arm = O.conv2d(arm, w[f"{p}/lateral_conv/conv/kernel"], None, 1, 1, "valid")
        offset = O.conv2d(torch.cat([arm, up * 2], dim=-1), w[f"{p}/offset_conv/kernel"], None, 1, 1, "valid")
        d = f"{p}/depack_l2"
        align = torch.relu(O.dcnv2(up, offset, w[f"{d}/kernel"], w[f"{d}/bias"], w[f"{d}/offset_kernel"], w[f"{d}/offset_bias"]))
        x = align + arm
        out.append(x)
    out.reverse()
    return out


def axial_attention_layer(w, prefix, x, heads, shared_qk=False):
    """layers/multihead_axial_attention.py:149-172"""
    def conv1x1(name, t):
        return O.conv2d(t, w[f"{prefix}/{name}/kernel"], w.get(f"{prefix}/{name}/bias"), 1, 1, "valid")

    q = conv1x1("query_conv", x)
    k = q if shared_qk else conv1x1("key_conv", x)
    return O.axial_attention_core(q, k, conv1x1("value_conv", x), heads)


# ------------------------------------------------------------------------------------------------------
# backbones/vit.py:19-63,66-113,163-183,277-323
# ------------------------------------------------------------------------------------------------------
def vit_forward(w, x, name, num_layer, pretrain_size, patch=16, dp_factors=None):
    """x [N,H,W,3] -> [N,H/16,W/16,C]; dp_factors[i] = (f_attn [N], f_mlp [N]) or None"""
    x = O.conv2d(x, w[f"{name}/patch_embed/projection/kernel"], w[f"{name}/patch_embed/projection/bias"], patch, 1, "same")
    N, H, W, C = x.shape
    x = x.reshape(N, H * W, C)
    x = torch.cat([w[f"{name}/class_token"].expand(N, 1, C), x], dim=1)
    pos = w[f"{name}/pos_embed"]
    g = pretrain_size // patch
    grid = O.resize_bicubic(pos[:, 1:].reshape(1, g, g, C), (H, W)).reshape(1, H * W, C)
    x = x + torch.cat([pos[:, :1], grid], dim=1)
    for i in range(num_layer):
        p = f"{name}/layers/{i}"
        y = O.layer_norm(x, w[f"{p}/ln1/gamma"], w[f"{p}/ln1/beta"], 1e-6)
        y = O.keras_mha_self(y, w[f"{p}/attn/query/kernel"], w[f"{p}/attn/query/bias"], w[f"{p}/attn/key/kernel"], w[f"{p}/attn/key/bias"],
                             w[f"{p}/attn/value/kernel"], w[f"{p}/attn/value/bias"], w[f"{p}/attn/attention_output/kernel"],
                             w[f"{p}/attn/attention_output/bias"])
        if dp_factors is not None and dp_factors[i] is not None:
            y = y * dp_factors[i][0].reshape(-1, 1, 1)
        x = ident = y + x
        y = O.layer_norm(x, w[f"{p}/ln2/gamma"], w[f"{p}/ln2/beta"], 1e-6)
        y = O.gelu(O.dense(y, w[f"{p}/ffn/dense0/kernel"], w[f"{p}/ffn/dense0/bias"]))
        y = O.dense(y, w[f"{p}/ffn/dense1/kernel"], w[f"{p}/ffn/dense1/bias"])
        if dp_factors is not None and dp_factors[i] is not None:
            y = y * dp_factors[i][1].reshape(-1, 1, 1)
        x = y + ident
    return x[:, 1:].reshape(N, H, W, C)


# ------------------------------------------------------------------------------------------------------
# backbones/eva/*: rotary table (rotar_embedding_cat.py:35-47,50-112,137-171), rot / apply_rot_embed_cat (:117-135), EvaAttention
# (attention.py:96-178), SwiGLU (swiglu.py:88-100), GluMlp (glumlp.py:96-112), Mlp (mlp.py), EvaBlock (block.py:140-183), Eva (eva.py:226-312)
# ------------------------------------------------------------------------------------------------------
def eva_rope_table(h, w, head_dim, temperature=10000.0):
    """RotaryEmbeddingCat(filters = head_dim, in_pixels=False, feat_shape=None, ref_feat_shape=None)([h, w]): float32 like the reference's tf ops,
    returned as float64 (sin [h w, head_dim], cos [h w, head_dim]).  numpy on purpose: the product builds its table with torch."""
    import numpy as np

    nb = head_dim // 4
    bands = (1.0 / (np.float32(temperature) ** (np.arange(nb, dtype=np.float32) / np.float32(nb)))).astype(np.float32)      # freq_bands(step=1)
    ty, tx = np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32)
    grid = np.stack(np.meshgrid(ty, tx, indexing="ij"), axis=-1)[..., None]      # [H, W, 2, 1]
    pos = (grid * bands).astype(np.float32)                                        # [H, W, 2, nb]
    sin = np.repeat(np.sin(pos).astype(np.float32).reshape(h * w, -1), 2, axis=-1)      # tf.repeat(repeats=[2], axis=-1)
    cos = np.repeat(np.cos(pos).astype(np.float32).reshape(h * w, -1), 2, axis=-1)
    return torch.from_numpy(sin).double(), torch.from_numpy(cos).double()


def eva_rot(x):
    return torch.stack([-x[..., 1::2], x[..., ::2]], dim=-1).reshape(x.shape)


def eva_attention(w, p, x, heads, fused, rope, prefix, use_norm):
    B, T, C = x.shape
    hd = C // heads
    if fused:
        qkv = O.dense(x, w[f"{p}/qkv/kernel"]) + torch.cat([w[f"{p}/q_bias"], torch.zeros_like(w[f"{p}/q_bias"]), w[f"{p}/v_bias"]])
        qkv = qkv.reshape(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
    else:
        q = O.dense(x, w[f"{p}/q_proj/kernel"], w.get(f"{p}/q_proj/bias")).reshape(B, T, heads, hd).permute(0, 2, 1, 3)
        k = O.dense(x, w[f"{p}/k_proj/kernel"]).reshape(B, T, heads, hd).permute(0, 2, 1, 3)
        v = O.dense(x, w[f"{p}/v_proj/kernel"], w.get(f"{p}/v_proj/bias")).reshape(B, T, heads, hd).permute(0, 2, 1, 3)
    if rope is not None:
        sin, cos = rope
        q = torch.cat([q[:, :, :prefix], q[:, :, prefix:] * cos + eva_rot(q[:, :, prefix:]) * sin], dim=2)
        k = torch.cat([k[:, :, :prefix], k[:, :, prefix:] * cos + eva_rot(k[:, :, prefix:]) * sin], dim=2)
    a = torch.softmax((q * hd ** -0.5) @ k.transpose(-1, -2), dim=-1)
    y = (a @ v).permute(0, 2, 1, 3).reshape(B, T, C)
    if use_norm:
        y = O.layer_norm(y, w[f"{p}/norm/gamma"], w[f"{p}/norm/beta"], 1e-6)
    return O.dense(y, w[f"{p}/proj/kernel"], w[f"{p}/proj/bias"])


def _eva_act(name):
    return {"gelu": O.gelu, "swish": lambda t: t * torch.sigmoid(t), "silu": lambda t: t * torch.sigmoid(t), "sigmoid": torch.sigmoid}[name]


def eva_mlp(w, p, x, kind, activation="gelu"):
    """kind: "swiglu" (SwiGLU with LayerNorm, gate activation = the block's activation), "glu" (GluMlp, swish gate on the second half, no norm),
    "mlp" / "mlp_norm" (Mlp without / with LayerNorm)"""
    if kind == "swiglu":
        y = _eva_act(activation)(O.dense(x, w[f"{p}/fc1_g/kernel"], w[f"{p}/fc1_g/bias"])) * O.dense(x, w[f"{p}/fc1_x/kernel"], w[f"{p}/fc1_x/bias"])
        y = O.layer_norm(y, w[f"{p}/norm/gamma"], w[f"{p}/norm/beta"], 1e-6)
    elif kind == "glu":
        y = O.dense(x, w[f"{p}/fc1/kernel"], w[f"{p}/fc1/bias"])
        h = y.shape[-1] // 2
        y = y[..., :h] * _eva_act("swish")(y[..., h:])
    else:
        y = _eva_act(activation)(O.dense(x, w[f"{p}/fc1/kernel"], w[f"{p}/fc1/bias"]))
        if kind == "mlp_norm":
            y = O.layer_norm(y, w[f"{p}/norm/gamma"], w[f"{p}/norm/beta"], 1e-6)
    return O.dense(y, w[f"{p}/fc2/kernel"], w[f"{p}/fc2/bias"])


def eva_block(w, p, x, heads, fused, rope, prefix, mlp_kind, scale_attention_inner=False, post_norm=False, activation="gelu", dp=None):
    residual = x
    y = x if post_norm else O.layer_norm(x, w[f"{p}/norm1/gamma"], w[f"{p}/norm1/beta"], 1e-6)
    y = eva_attention(w, f"{p}/attn", y, heads, fused, rope, prefix, scale_attention_inner)
    if post_norm:
        y = O.layer_norm(y, w[f"{p}/norm1/gamma"], w[f"{p}/norm1/beta"], 1e-6)
    if f"{p}/gamma_1" in w:
        y = y * w[f"{p}/gamma_1"]
    if dp is not None:
        y = y * dp.reshape(-1, 1, 1)
    x = residual = y + residual
    y = x if post_norm else O.layer_norm(x, w[f"{p}/norm2/gamma"], w[f"{p}/norm2/beta"], 1e-6)
    y = eva_mlp(w, f"{p}/mlp", y, mlp_kind, activation)
    if post_norm:
        y = O.layer_norm(y, w[f"{p}/norm2/gamma"], w[f"{p}/norm2/beta"], 1e-6)
    if f"{p}/gamma_2" in w:
        y = y * w[f"{p}/gamma_2"]
    return y + residual      # (block.py:180-181: no drop path on the second branch)


def eva_forward(w, x, name, depth, heads, patch, fused, mlp_kind, grid_size, scale_attention_inner=False, post_norm=False, activation="gelu",
                dp_factors=None):
    """Eva.call with return_endpoints=True: [class_token, patch_embedding, block outputs ...]; grid_size = the token grid the position embedding
    was built for (it is resampled bilinearly to this call's grid)"""
    x = O.conv2d(x, w[f"{name}/patch_embed/projection/kernel"], w[f"{name}/patch_embed/projection/bias"], patch, 1, "valid")
    patch_embedding = x
    N, H, W, C = x.shape
    sin, cos = eva_rope_table(H, W, C // heads)
    x = torch.cat([w[f"{name}/class_token"].expand(N, 1, C), x.reshape(N, H * W, C)], dim=1)
    pos = w[f"{name}/pos_embed"]
    grid = O.resize_bilinear(pos[:, 1:].reshape(1, grid_size[0], grid_size[1], C), (H, W)).reshape(1, H * W, C)
    x = x + torch.cat([pos[:, :1], grid], dim=1)
    ends = []
    for i in range(depth):
        x = eva_block(w, f"{name}/blocks/{i}", x, heads, fused, (sin, cos), 1, mlp_kind, scale_attention_inner, post_norm, activation,
                      None if dp_factors is None else dp_factors[i])
        ends.append(x[:, 1:].reshape(N, H, W, C))
    return [x[:, :1], patch_embedding] + ends


# ------------------------------------------------------------------------------------------------------
# backbones/swin.py: window_partition/reverse :46-64, WindowAttention :117-167, block :237-294, PatchMerging :309-337,
# mask :391-433, PatchEmbed :479-501, model :601-622
# ------------------------------------------------------------------------------------------------------
def _window_partition(x, ws):
    B, H, W, C = x.shape
    x = x.reshape(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, ws, ws, C)


def _window_reverse(win, ws, H, W, C):
    x = win.reshape(-1, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, H, W, C)


def swin_attention_mask(H, W, ws, shift, dtype=torch.float64):
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    parts, cnt = [], 0
    for hl in (Hp - ws, ws - shift, shift):
        row = []
        for wl in (Wp - ws, ws - shift, shift):
            row.append(torch.full((1, hl, wl, 1), float(cnt), dtype=dtype))
            cnt += 1
        parts.append(torch.cat(row, dim=2))
    img = torch.cat(parts, dim=1)
    mw = _window_partition(img, ws).reshape(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return torch.where(m != 0, torch.full_like(m, -100.0), torch.zeros_like(m))


def _rel_index(ws):
    import numpy as np

    coords = np.stack(np.meshgrid(np.arange(ws), np.arange(ws), indexing="ij")).reshape(2, -1)
    rel = (coords[:, :, None] - coords[:, None, :]).transpose(1, 2, 0).copy()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return torch.from_numpy(rel.sum(-1).astype(np.int64))


def swin_window_attention(w, p, x, heads, ws, mask):
    B_, N, C = x.shape
    qkv = O.dense(x, w[f"{p}/qkv/kernel"], w[f"{p}/qkv/bias"]).reshape(B_, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (C // heads) ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-1, -2)
    bias = w[f"{p}/relative_position_bias_table"][_rel_index(ws).reshape(-1)].reshape(N, N, heads).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.reshape(-1, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).reshape(-1, heads, N, N)
    attn = torch.softmax(attn, dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    return O.dense(x, w[f"{p}/proj/kernel"], w[f"{p}/proj/bias"])


def swin_block(w, p, x, heads, ws, shift, mask, dp=None):
    import torch.nn.functional as TF

    B, H, W, C = x.shape
    shortcut = x.reshape(B, H * W, C)
    y = O.layer_norm(x, w[f"{p}/norm1/gamma"], w[f"{p}/norm1/beta"], 1e-5)
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    y = TF.pad(y, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    if shift > 0:
        y = torch.roll(y, (-shift, -shift), dims=(1, 2))
    win = _window_partition(y, ws).reshape(-1, ws * ws, C)
    win = swin_window_attention(w, f"{p}/attn", win, heads, ws, mask if shift > 0 else None)
    y = _window_reverse(win.reshape(-1, ws, ws, C), ws, Hp, Wp, C)
    if shift > 0:
        y = torch.roll(y, (shift, shift), dims=(1, 2))
    y = y[:, :H, :W].reshape(B, H * W, C)
    if dp is not None:
        y = y * dp[0].reshape(-1, 1, 1)
    x = shortcut + y
    y = O.layer_norm(x, w[f"{p}/norm2/gamma"], w[f"{p}/norm2/beta"], 1e-5)
    y = O.dense(O.gelu(O.dense(y, w[f"{p}/mlp/fc1/kernel"], w[f"{p}/mlp/fc1/bias"])), w[f"{p}/mlp/fc2/kernel"], w[f"{p}/mlp/fc2/bias"])
    if dp is not None:
        y = y * dp[1].reshape(-1, 1, 1)
    return (x + y).reshape(B, H, W, C)


def swin_patch_merging(w, p, x):
    import torch.nn.functional as TF

    B, H, W, C = x.shape
    x = TF.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], dim=-1)
    H2, W2 = x.shape[1], x.shape[2]
    x = O.layer_norm(x.reshape(B, H2 * W2, 4 * C), w[f"{p}/norm/gamma"], w[f"{p}/norm/beta"], 1e-5)
    return O.dense(x, w[f"{p}/reduction/kernel"]).reshape(B, H2, W2, 2 * C)


def swin_forward(w, x, depths=(2, 2, 6, 2), heads=(3, 6, 12, 24), ws=7, patch=4, dp_factors=None, ape=None):
    """endpoints [patch_embed, l0, l1, l2, l3] (pre-downsample).  dp_factors[layer][block] = (f_attn, f_mlp) or None; ape = name of the
    absolute position embedding [1, patches, C] added to the patch embedding (backbones/swin.py:563-569,606-607) or None"""
    import torch.nn.functional as TF

    H, W = x.shape[1], x.shape[2]
    x = TF.pad(x, (0, 0, 0, (patch - W % patch) % patch, 0, (patch - H % patch) % patch))
    x = O.conv2d(x, w["patch_embed/proj/kernel"], w["patch_embed/proj/bias"], patch, 1, "valid")
    x = O.layer_norm(x, w["patch_embed/norm/gamma"], w["patch_embed/norm/beta"], 1e-5)
    if ape is not None:
        x = x + w[ape].reshape(1, *x.shape[1:])      # (the reference reshapes to shape(x): batch 1 only; the batch broadcast is the evident intent)
    endpoints = [x]
    for li, depth in enumerate(depths):
        mask = swin_attention_mask(x.shape[1], x.shape[2], ws, ws // 2, x.dtype)
        for bi in range(depth):
            dp = None if dp_factors is None else dp_factors[li][bi]
            x = swin_block(w, f"layers/{li}/blocks/{bi}", x, heads[li], ws, 0 if bi % 2 == 0 else ws // 2, mask, dp)
        endpoints.append(x)
        if li < len(depths) - 1:
            x = swin_patch_merging(w, f"layers/{li}/downsample", x)
    return endpoints


# ------------------------------------------------------------------------------------------------------
# layers/dcn_v3/dcn_v3.py:107-150 and backbones/intern_image/*
# ------------------------------------------------------------------------------------------------------
def dcnv3_layer(w, p, x, groups, kernel_size=3, dw_kernel=None, offset_scale=1.0):
    N, H, W, C = x.shape
    x_proj = O.dense(x, w[f"{p}/input_proj/kernel"], w[f"{p}/input_proj/bias"])
    x1 = O.depthwise_conv2d(x, w[f"{p}/dw_conv/depthwise_kernel"], w[f"{p}/dw_conv/bias"], 1, 1, "same")
    x1 = O.gelu(O.layer_norm(x1, w[f"{p}/dw_conv_norm/gamma"], w[f"{p}/dw_conv_norm/beta"], 1e-6))
    offset = O.dense(x1, w[f"{p}/offset/kernel"], w[f"{p}/offset/bias"])
    mask = O.dense(x1, w[f"{p}/mask/kernel"], w[f"{p}/mask/bias"])
    mask = torch.softmax(mask.reshape(N, H, W, groups, -1), dim=-1).reshape(N, H, W, -1)
    y = O.dcnv3_op(x_proj, offset, mask, (kernel_size, kernel_size), (1, 1), "SAME", (1, 1), groups, C // groups, offset_scale)
    if f"{p}/center_feature_scale_proj/kernel" in w:      # layers/dcn_v3/dcn_v3.py:138-146 (no sigmoid in this port)
        s = O.dense(x1, w[f"{p}/center_feature_scale_proj/kernel"], w[f"{p}/center_feature_scale_proj/bias"])
        s = s.unsqueeze(-1).expand(N, H, W, groups, C // groups).reshape(N, H, W, C)
        y = y * (1 - s) + x_proj * s
    return O.dense(y, w[f"{p}/output_proj/kernel"], w[f"{p}/output_proj/bias"])


def intern_image_layer(w, p, x, groups, post_norm, dp=None):
    def ln(name, t):
        return O.layer_norm(t, w[f"{p}/{name}/gamma"], w[f"{p}/{name}/beta"], 1e-6)

    def mlp(t):
        t = O.gelu(O.dense(t, w[f"{p}/mlp/fc1/kernel"], w[f"{p}/mlp/fc1/bias"]))
        return O.dense(t, w[f"{p}/mlp/fc2/kernel"], w[f"{p}/mlp/fc2/bias"])

    def drop(t, i):
        return t if dp is None else t * dp[i].reshape(-1, 1, 1, 1)

    residual = x
    if post_norm:
        y = drop(ln("norm1", dcnv3_layer(w, f"{p}/dcn", x, groups)) * w[f"{p}/gamma1"], 0)
        residual = x = residual + y
        y = drop(ln("norm2", mlp(x)) * w[f"{p}/gamma2"], 1)
        return y + residual
    y = drop(dcnv3_layer(w, f"{p}/dcn", ln("norm1", x), groups) * w[f"{p}/gamma1"], 0)
    residual = x = residual + y
    y = drop(mlp(ln("norm2", x)) * w[f"{p}/gamma2"], 1)
    return y + residual


def intern_image_forward(w, x, depths, groups, post_norm, dp_factors=None):
    """endpoints [stem before 2nd stride, b0, b1, ...] (pre-downsample); the centre-feature scale of the DCNv3 layers is on when its
    projection is among the weights (it also keeps the block norm under post-norm: intern_image_block.py:81,115)"""
    center = any(k.endswith("center_feature_scale_proj/kernel") for k in w)
    def ln(p, t):
        return O.layer_norm(t, w[f"{p}/gamma"], w[f"{p}/beta"], 1e-6)

    x = O.gelu(ln("patch_embed/norm1", O.conv2d(x, w["patch_embed/conv1/kernel"], w["patch_embed/conv1/bias"], 2, 1, "same")))
    endpoints = [x]
    x = ln("patch_embed/norm2", O.conv2d(x, w["patch_embed/conv2/kernel"], w["patch_embed/conv2/bias"], 2, 1, "same"))
    for bi, depth in enumerate(depths):
        for li in range(depth):
            dp = None if dp_factors is None else dp_factors[bi][li]
            x = intern_image_layer(w, f"block/{bi}/layer/{li}", x, groups[bi], post_norm, dp)
        if not post_norm or center:
            x = ln(f"block/{bi}/norm", x)
        endpoints.append(x)
        if bi < len(depths) - 1:
            x = ln(f"block/{bi}/downsample/norm", O.conv2d(x, w[f"block/{bi}/downsample/conv/kernel"], None, 2, 1, "same"))
    return endpoints


# ------------------------------------------------------------------------------------------------------
# model-level compositions of iseg_amd/heads.py and the inference drivers of core_inference.py:230-304, core_model.py:170-326
# ------------------------------------------------------------------------------------------------------
def resnet_aspp_forward(w, x, training=False, output_stride=32, head="aspp_head", seg="seg", new_stats=None):
    ends = resnet_forward(w, x, output_stride=output_stride, training=training, new_stats=new_stats)
    mult = max(32 // output_stride, 1)
    feat = aspp(w, f"{head}/aspp", ends[-1], training, rates=tuple(r * mult for r in (3, 6, 9)), new_stats=new_stats)
    feat = conv_norm_act(w, f"{head}/end_conv", feat, training, new_stats=new_stats)
    small = O.conv2d(feat, w[f"{seg}/logits_conv/kernel"], w[f"{seg}/logits_conv/bias"], 1, 1, "same")
    return {"endpoints": ends, "logits": O.resize_bilinear(small, (x.shape[1], x.shape[2]))}


def swin_fpn_forward(w, x, training=False, depths=(2, 2, 6, 2), heads=(3, 6, 12, 24), ws=7, head="fpn_head", seg="seg", dp_factors=None,
                     new_stats=None):
    """BASELINE config 3's composition (iseg_amd/heads.py FPNHead): Swin (backbones/swin.py:601-635, endpoints [patch_embed, l0 .. l3]) ->
    FeaturePyramidNetwork(skip_conv_filters = C_top)(endpoints[1:]) (layers/fpn.py:16-61) -> finest level -> ConvNormAct(256, 1x1) -> logits
    1x1 (layers/core_model_ext.py:185-196) -> bilinear to the input size"""
    ends = swin_forward(w, x, depths=depths, heads=heads, ws=ws, dp_factors=dp_factors)
    levels = fpn(w, f"{head}/fpn", ends[1:], training, new_stats=new_stats)
    feat = conv_norm_act(w, f"{head}/end_conv", levels[0], training, new_stats=new_stats)
    small = O.conv2d(feat, w[f"{seg}/logits_conv/kernel"], w[f"{seg}/logits_conv/bias"], 1, 1, "same")
    return {"endpoints": ends, "levels": levels, "head": feat, "logits": O.resize_bilinear(small, (x.shape[1], x.shape[2]))}


def intern_image_aspp_forward(w, x, training=False, depths=(4, 4, 21, 4), groups=(7, 14, 28, 56), post_norm=True, head="aspp_head", seg="seg",
                              dp_factors=None, new_stats=None):
    """BASELINE config 5's composition: InternImage (backbones/intern_image/intern_image.py:119-135; the "base" sizes, which the reference
    leaves to register_backbone) -> ASPP on the last endpoint (layers/aspp.py:7-71, rates 3/6/9 at output stride 32) -> ConvNormAct(256, 1x1)
    -> logits 1x1 -> bilinear to the input size"""
    ends = intern_image_forward(w, x, depths, groups, post_norm, dp_factors)
    feat = aspp(w, f"{head}/aspp", ends[-1], training, rates=(3, 6, 9), new_stats=new_stats)
    feat = conv_norm_act(w, f"{head}/end_conv", feat, training, new_stats=new_stats)
    small = O.conv2d(feat, w[f"{seg}/logits_conv/kernel"], w[f"{seg}/logits_conv/bias"], 1, 1, "same")
    return {"endpoints": ends, "head": feat, "logits": O.resize_bilinear(small, (x.shape[1], x.shape[2]))}


def vit_simple_decoder_forward(w, x, training=False, backbone="ViT-B_16", num_layer=12, pretrain_size=384, head="decoder_head", seg="seg"):
    """BASELINE config 4's composition: ViT (backbones/vit.py:277-323, one endpoint) -> ConvNormAct(256, 1x1) as the high-level
    branch -> SimpleDecoder(48, 256) with the endpoint itself as low-level input (layers/simpledecoder.py:21-36) -> logits 1x1 -> bilinear"""
    e = vit_forward(w, x, backbone, num_layer, pretrain_size)
    high = conv_norm_act(w, f"{head}/high_conv", e, training)
    feat = simple_decoder(w, f"{head}/decoder", e, high, training)
    small = O.conv2d(feat, w[f"{seg}/logits_conv/kernel"], w[f"{seg}/logits_conv/bias"], 1, 1, "same")
    return {"endpoint": e, "logits": O.resize_bilinear(small, (x.shape[1], x.shape[2]))}


def sliding_window_inference(fn, x, window):
    """core_inference.py:230-304: windows at get_sliding_start_indexs positions, logits zero-padded back and summed, divided by
    the per-pixel visit count"""
    N, H, W, _ = x.shape
    wh, ww = min(window[0], H), min(window[1], W)
    total, count = None, torch.zeros(H, W, dtype=x.dtype)
    for t in O.sliding_start_indexs(H, wh):
        for l in O.sliding_start_indexs(W, ww):
            logits = fn(x[:, t:t + wh, l:l + ww])
            if total is None:
                total = torch.zeros(N, H, W, logits.shape[-1], dtype=x.dtype)
            total[:, t:t + wh, l:l + ww] += logits
            count[t:t + wh, l:l + ww] += 1
    return total / count.reshape(1, H, W, 1)


def multi_scale_inference(fn, x, scale_rates=(1.0,), flip=False):
    """core_model.py:231-326 with get_scaled_size(pad_mode=1) (utils/common.py:159-188)"""
    def scaled(h, rate):
        t = int(rate * h)
        return t + 1 if (t % 2 == 0 and h % 2 != 0) else t

    H, W = x.shape[1], x.shape[2]
    total = 0
    for mirrored in ([False, True] if flip else [False]):
        xi = torch.flip(x, dims=(2,)) if mirrored else x
        part = 0
        for r in scale_rates:
            xs = O.resize_bilinear(xi, (scaled(H, r), scaled(W, r)))
            part = part + O.resize_bilinear(fn(xs), (H, W))
        total = total + (torch.flip(part, dims=(2,)) if mirrored else part)
    return total / (len(scale_rates) * (2 if flip else 1))


# ------------------------------------------------------------------------------------------------------
# N optimisation steps of CoreTrain's hot loop (core_train.py:141-152) restated on the oracle: forward (batch statistics) ->
# mean ignore-label CE -> autograd backward -> Keras AdamW (optimizers/modern/adamw.py:13-74) with the poly-decay schedule ->
# moving-statistics update.  Used by smoke() and tests/test_model_gpu.py as the loss-curve checker.
# ------------------------------------------------------------------------------------------------------
class ConvNeXtASPPAdamWSteps:
    """The hot loop one step at a time, so that a checker can hand each step's gradient MASK to the implementation under test.

    Adam (adamw.py:13-59) at Keras' default epsilon 1e-7 is scale-free: an element whose true gradient lies below the fp32 rounding noise of
    the sums that produce it moves by +-lr with a sign nobody owns (at step 1 every element moves by exactly lr * sign(g)).  `tau` zeroes the
    gradient elements below tau * (largest gradient element of the model, fp64) -- on this side, and through `masks` on the other side --
    which removes exactly that ill-conditioned part of the comparison: measured on the smoke problem, the 5-step curves of the HIP path
    and of this restatement differ by 1.8e-2 (tau = 0), 5.6e-3 (1e-6), 7.5e-4 (1e-5), 3.5e-5 (1e-4), 5.7e-7 (1e-3)."""

    def __init__(self, w, x, y, trainable, lr_fn, wd_of, eps=1e-7, beta1=0.9, beta2=0.999, output_stride=32, tau=0.0):
        self.w, self.x, self.y, self.trainable = w, x, y, list(trainable)
        self.lr_fn, self.wd_of, self.eps, self.b1, self.b2, self.os, self.tau = lr_fn, wd_of, eps, beta1, beta2, output_stride, tau
        self.state = {k: (torch.zeros_like(w[k]), torch.zeros_like(w[k])) for k in self.trainable}
        self.step_index = 0
        self._pending = None

    def forward_backward(self):
        """loss of the current weights and, for tau > 0, {name: bool mask of the gradient elements that take part in the update}"""
        w = self.w
        wr = {k: (v.clone().requires_grad_(True) if k in self.state else v) for k, v in w.items()}
        new_stats = {}
        out = convnext_aspp_forward(wr, self.x.to(next(iter(w.values())).dtype), training=True, output_stride=self.os, new_stats=new_stats)
        loss = mean_ce_loss(out["logits"], self.y)
        loss.backward()
        grads = {k: wr[k].grad for k in self.trainable}
        masks = None
        if self.tau > 0:
            top = max(g.abs().max().item() for g in grads.values())
            masks = {k: g.abs() >= self.tau * top for k, g in grads.items()}
            grads = {k: g * masks[k] for k, g in grads.items()}
        self._pending = (grads, new_stats)
        return loss.item(), masks

    def apply(self):
        grads, new_stats = self._pending
        lr = self.lr_fn(self.step_index)
        for k in self.trainable:
            m, v = self.state[k]
            nw, nm, nv = O.adamw_step(self.w[k], grads[k], m, v, self.step_index + 1, lr, 1.0, self.wd_of(k), beta1=self.b1, beta2=self.b2,
                                      eps=self.eps)
            self.w[k], self.state[k] = nw.detach(), (nm, nv)
        self.w.update(new_stats)
        self.step_index += 1
        self._pending = None


class ConvNeXtASPPSGDSteps(ConvNeXtASPPAdamWSteps):
    """The same loop under SGD_EXT with momentum (optimizers/modern/sgd.py:38-51): m = -g lr + m mu; w += m.  No scale-free division, so nothing
    amplifies rounding noise and the comparison needs no gradient mask: any residual against this curve is a real gradient difference."""

    def __init__(self, w, x, y, trainable, lr_fn, momentum=0.9, l2_of=None, output_stride=32):
        super().__init__(w, x, y, trainable, lr_fn, lambda k: 0.0, output_stride=output_stride, tau=0.0)
        self.momentum, self.l2_of = momentum, (l2_of or (lambda k: 0.0))
        self.state = {k: torch.zeros_like(w[k]) for k in self.trainable}
        self.last_grads = None

    def apply(self):
        grads, new_stats = self._pending
        self.last_grads = grads
        lr = self.lr_fn(self.step_index)
        for k in self.trainable:
            nw, nm = O.sgd_step(self.w[k], grads[k], self.state[k], lr, 1.0, self.momentum, self.l2_of(k))
            self.w[k], self.state[k] = nw.detach(), nm
        self.w.update(new_stats)
        self.step_index += 1
        self._pending = None


def convnext_aspp_adamw_curve(w, x, y, steps, trainable, lr_fn, wd_of, eps=1e-7, beta1=0.9, beta2=0.999, output_stride=32):
    """w: weight dict (updated in place); trainable: names of the optimised variables; lr_fn(step) -> lr; wd_of(name) -> weight decay.
    Returns the list of losses, one per step (the loss BEFORE that step's update, like Keras logs it)."""
    run = ConvNeXtASPPAdamWSteps(w, x, y, trainable, lr_fn, wd_of, eps, beta1, beta2, output_stride)
    curve = []
    for _ in range(steps):
        curve.append(run.forward_backward()[0])
        run.apply()
    return curve


# ------------------------------------------------------------------------------------------------------
# HRNet (backbones/hrnet.py): Bottleneck stem layer, transition / branch / fuse modules per stage, aligned-corner resizes.
# The fuse module writes each fused branch back into the list it reads from (:287-309), so branch i > 0 is built from the already
# fused lower branches -- restated literally.
# ------------------------------------------------------------------------------------------------------
def _hr_conv_bn(w, conv, bn, x, training, stride=1, relu=True, new_stats=None):
    y = O.conv2d(x, w[f"{conv}/kernel"], None, stride, 1, "same")
    y = _bn(w, bn, y, training, 1e-3, new_stats=new_stats)
    return torch.relu(y) if relu else y


def _hr_block(w, name, x, bottleneck, downsample, training, new_stats):
    residual = x
    if downsample:      # DownSampleBlock :157-173 (1x1 conv, no activation)
        residual = _hr_conv_bn(w, f"{name}/downsample/0", f"{name}/downsample/1", x, training, relu=False, new_stats=new_stats)
    y = _hr_conv_bn(w, f"{name}/conv1", f"{name}/bn1", x, training, new_stats=new_stats)
    if bottleneck:
        y = _hr_conv_bn(w, f"{name}/conv2", f"{name}/bn2", y, training, new_stats=new_stats)
        y = _hr_conv_bn(w, f"{name}/conv3", f"{name}/bn3", y, training, relu=False, new_stats=new_stats)
    else:
        y = _hr_conv_bn(w, f"{name}/conv2", f"{name}/bn2", y, training, relu=False, new_stats=new_stats)
    return torch.relu(y + residual)


def _hr_layer(w, name, x, bottleneck, filters, num_blocks, training, new_stats):
    expansion = 4 if bottleneck else 1
    for i in range(num_blocks):
        x = _hr_block(w, f"{name}/{i}", x, bottleneck, i == 0 and x.shape[-1] != filters * expansion, training, new_stats)
    return x


def _hr_conv_block(w, name, x, training, stride=1, relu=True, new_stats=None):
    return _hr_conv_bn(w, f"{name}/0", f"{name}/1", x, training, stride, relu, new_stats)


def _hr_transition(w, name, x_list, filters_list, training, new_stats):
    num_in = len(x_list)
    out = []
    for i, f in enumerate(filters_list):
        if i < num_in:
            out.append(_hr_conv_block(w, f"{name}/{i}", x_list[i], training, new_stats=new_stats) if f != x_list[i].shape[-1] else x_list[i])
        else:
            y = x_list[-1]
            for j in range(i + 1 - num_in):
                y = _hr_conv_block(w, f"{name}/{i}/{j}", y, training, stride=2, new_stats=new_stats)
            out.append(y)
    return out


def _hr_fuse(w, name, x_list, training, new_stats):
    x_list = list(x_list)
    nb = len(x_list)
    for i in range(nb):
        if i == 0:
            y = x_list[0]
        else:
            y = x_list[0]
            for k in range(i):      # HighResolutionFuseStack(i, 0): i stride-2 blocks, ReLU on all but the last
                y = _hr_conv_block(w, f"{name}/{i}/0/{k}", y, training, stride=2, relu=k != i - 1, new_stats=new_stats)
        for j in range(1, nb):
            x = x_list[j]
            if j > i:
                x = _hr_conv_block(w, f"{name}/{i}/{j}", x, training, relu=False, new_stats=new_stats)
                x = O.resize_bilinear(x, x_list[i].shape[1:3], align_corners=True)
            elif j < i:
                for k in range(i - j):
                    x = _hr_conv_block(w, f"{name}/{i}/{j}/{k}", x, training, stride=2, relu=k != i - j - 1, new_stats=new_stats)
            y = y + x
        x_list[i] = torch.relu(y)
    return x_list


def hrnet_forward(w, x, stages, training=False, new_stats=None):
    """stages: [(num_modules, filters_list, num_blocks_list), ...] as HighResolutionNet.add_stage receives them (BasicBlock stages);
    returns the branch list followed by the concatenation at the highest resolution (:506-538, return_endpoints=True)"""
    x = _hr_conv_bn(w, "conv1", "bn1", x, training, stride=2, new_stats=new_stats)
    x = _hr_conv_bn(w, "conv2", "bn2", x, training, stride=2, new_stats=new_stats)
    x = _hr_layer(w, "layer1", x, True, 64, 4, training, new_stats)
    x_list = [x]
    for s, (num_modules, filters_list, num_blocks_list) in enumerate(stages):
        name = f"stage{s + 2}"
        x_list = _hr_transition(w, f"{name}/transition", x_list, filters_list, training, new_stats)
        for m in range(num_modules):
            x_list = [_hr_layer(w, f"{name}/{m}/branches/{i}", x_list[i], False, filters_list[i], num_blocks_list[i], training, new_stats)
                      for i in range(len(x_list))]
            if len(x_list) > 1:
                x_list = _hr_fuse(w, f"{name}/{m}/fuse_layers", x_list, training, new_stats)
    size = x_list[0].shape[1:3]
    y = torch.cat([x_list[0]] + [O.resize_bilinear(t, size, align_corners=True) for t in x_list[1:]], dim=-1)
    return x_list + [y]


# ------------------------------------------------------------------------------------------------------
# MobileNetV2 (backbones/mobilenetv2_common.py): stride-2 blocks are restated literally -- ZeroPadding2D(correct_pad) followed by the
# depthwise convolution with padding "valid" (:114-176) -- so that the product's 'same' stride-2 shortcut is checked against it.
# ------------------------------------------------------------------------------------------------------
MOBILENETV2_BLOCKS = [(16, 1, 1, 1), (24, 2, 6, 2), (32, 2, 6, 3), (64, 2, 6, 4), (96, 1, 6, 3), (160, 2, 6, 3), (320, 1, 6, 1)]      # filters, stride, expansion, repeats


def mobilenetv2_forward(w, x, output_stride=32, training=False, new_stats=None):
    """returns the endpoint list of MobileNetV2.call(return_endpoints=True) after build_atrous_mobilenetv2(output_stride)"""
    def bn(name, y):
        return _bn(w, name, y, training, 1e-3, new_stats=new_stats)

    relu6 = lambda t: torch.clamp(t, 0.0, 6.0)      # noqa: E731
    x = relu6(bn("bn_Conv1", O.conv2d(x, w["Conv1/kernel"], None, 2, 1, "same")))
    endpoints = []
    block_id, current_os, rate = 0, 2, 1
    for filters, stride0, expansion, repeats in MOBILENETV2_BLOCKS:
        for r in range(repeats):
            original_stride = stride0 if r == 0 else 1
            stride, dil = original_stride, 1
            if original_stride > 1:      # build_atrous_mobilenetv2 :204-222
                if current_os >= output_stride:
                    rate *= original_stride
                    stride, dil = 1, rate
                else:
                    current_os *= original_stride
            else:
                dil = rate
            prefix = f"block_{block_id}_" if block_id else "expanded_conv_"
            if original_stride > 1:
                endpoints.append(x)
            inp = x
            y = x
            if block_id:
                y = relu6(bn(f"{prefix}expand_BN", O.conv2d(y, w[f"{prefix}expand/kernel"], None, 1, 1, "same")))
            if stride == 2:
                H, W = y.shape[1], y.shape[2]
                pt, pl = 1 - (1 - H % 2), 1 - (1 - W % 2)      # correct_pad(., 3): (1 - adjust, 1) per axis
                y = torch.nn.functional.pad(y, (0, 0, pl, 1, pt, 1))
                y = O.depthwise_conv2d(y, w[f"{prefix}depthwise/depthwise_kernel"], None, 2, 1, "valid")
            else:
                y = O.depthwise_conv2d(y, w[f"{prefix}depthwise/depthwise_kernel"], None, 1, dil, "same")
            y = relu6(bn(f"{prefix}depthwise_BN", y))
            y = bn(f"{prefix}project_BN", O.conv2d(y, w[f"{prefix}project/kernel"], None, 1, 1, "same"))
            x = inp + y if (inp.shape[-1] == y.shape[-1] and original_stride == 1) else y
            block_id += 1
    x = relu6(bn("Conv_1_bn", O.conv2d(x, w["Conv_1/kernel"], None, 1, 1, "same")))
    return endpoints + [x]
