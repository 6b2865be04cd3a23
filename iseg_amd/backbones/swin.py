"""backbones/swin.py of the reference: Mlp (:17-43), window_partition / window_reverse (:46-64), WindowAttention (:67-167),
SwinTransformerBlock (:180-294), PatchMerging (:297-337), BasicLayer (:340-455), PatchEmbed (:458-501),
SwinTransformerModel (:504-622), swin_tiny_224 / swin_base_384 / swin_large_384 (:625-662).

MI355X mapping: zero-pad + cyclic roll + window partition (and their inverses + crop) are ONE row gather each, driven by int32
index tables built on the host from the static geometry; PatchMerging's 2x2 space-to-depth is another; the window attention is
functional.attention_packed with the relative-position bias table gather and the 0 / -100 shift mask folded into the softmax."""
import numpy as np
import torch

from .. import functional as F
from .. import nn
from ..layers.base_layers import Conv2D, Dense, Dropout, LayerNormalization
from ..nn import Layer

_INDEX_CACHE = {}


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(nn.device())


def window_index_tables(N, H, W, ws, shift):
    """(partition, reverse, Hp, Wp): partition[r] = source row in [N*H*W] of window-token r (-1 = zero padding),
    reverse[n*H*W + h*W + w] = window-token row that lands on (n, h, w) after un-rolling and cropping."""
    key = ("win", N, H, W, ws, shift, str(nn.device()))
    if key not in _INDEX_CACHE:
        Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
        nWh, nWw = Hp // ws, Wp // ws
        # padded, shifted coordinates of every window token
        wh, ww, i, j = np.meshgrid(np.arange(nWh), np.arange(nWw), np.arange(ws), np.arange(ws), indexing="ij")
        sh = (wh * ws + i + shift) % Hp           # shifted_x = tf.roll(x, -shift): shifted[p] = x[(p + shift) % Hp]
        sw = (ww * ws + j + shift) % Wp
        src = np.where((sh < H) & (sw < W), sh * W + sw, -1).reshape(-1)
        part = np.concatenate([np.where(src >= 0, src + n * H * W, -1) for n in range(N)]).astype(np.int32)
        h, w = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        ph, pw = (h - shift) % Hp, (w - shift) % Wp   # x = tf.roll(shifted, +shift): x[p] = shifted[(p - shift) % Hp]
        row = ((ph // ws) * nWw + (pw // ws)) * ws * ws + (ph % ws) * ws + (pw % ws)
        rev = np.concatenate([row.reshape(-1) + n * nWh * nWw * ws * ws for n in range(N)]).astype(np.int32)
        _INDEX_CACHE[key] = (_dev(part), _dev(rev), Hp, Wp)
    return _INDEX_CACHE[key]


def merge_index_tables(N, H, W):
    """PatchMerging (:316-327): pad to even, x0 = [0::2, 0::2], x1 = [1::2, 0::2], x2 = [0::2, 1::2], x3 = [1::2, 1::2]"""
    key = ("merge", N, H, W, str(nn.device()))
    if key not in _INDEX_CACHE:
        H2, W2 = (H + 1) // 2, (W + 1) // 2
        h2, w2, q = np.meshgrid(np.arange(H2), np.arange(W2), np.arange(4), indexing="ij")
        sh, sw = 2 * h2 + (q % 2), 2 * w2 + (q // 2)
        src = np.where((sh < H) & (sw < W), sh * W + sw, -1).reshape(-1)
        fwd = np.concatenate([np.where(src >= 0, src + n * H * W, -1) for n in range(N)]).astype(np.int32)
        h, w = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        row = ((h // 2) * W2 + (w // 2)) * 4 + (h % 2) + 2 * (w % 2)
        bwd = np.concatenate([row.reshape(-1) + n * H2 * W2 * 4 for n in range(N)]).astype(np.int32)
        _INDEX_CACHE[key] = (_dev(fwd), _dev(bwd), H2, W2)
    return _INDEX_CACHE[key]


def relative_position_index(ws):
    """:93-104"""
    coords = np.stack(np.meshgrid(np.arange(ws[0]), np.arange(ws[1]), indexing="ij")).reshape(2, -1)
    rel = (coords[:, :, None] - coords[:, None, :]).transpose(1, 2, 0).copy()
    rel[:, :, 0] += ws[0] - 1
    rel[:, :, 1] += ws[1] - 1
    rel[:, :, 0] *= 2 * ws[1] - 1
    return rel.sum(-1).astype(np.int32)


def shift_attention_mask(H, W, ws, shift):
    """BasicLayer.generate_attention_mask (:391-433): regions of the padded, shifted image; 0 inside a region, -100 across"""
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    img = np.zeros((Hp, Wp), dtype=np.float32)
    cnt = 0
    h0 = 0
    for hl in (Hp - ws, ws - shift, shift):
        w0 = 0
        for wl in (Wp - ws, ws - shift, shift):
            img[h0:h0 + hl, w0:w0 + wl] = cnt
            cnt += 1
            w0 += wl
        h0 += hl
    mw = img.reshape(Hp // ws, ws, Wp // ws, ws).transpose(0, 2, 1, 3).reshape(-1, ws * ws)
    diff = mw[:, None, :] - mw[:, :, None]
    return np.where(diff != 0, -100.0, 0.0).astype(np.float32)


class Mlp(Layer):
    def __init__(self, in_features, hidden_features=None, out_features=None, dropout_rate=0.0, name=None):
        super().__init__(name=name)
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = Dense(hidden_features, activation="gelu", name=f"{name}/fc1")
        self.fc2 = Dense(out_features, name=f"{name}/fc2")
        self.dropout = Dropout(dropout_rate, name=f"{name}/dropout")

    def fusable(self, training):
        return self.fc1.built and self.fc2.built and (self.dropout.rate == 0.0 or not training)

    def call(self, inputs, training=None, residual=None, drop_path_mask=None):
        """residual / drop_path_mask (fusable(training) only): residual + factor[sample] * mlp(inputs) from the second product's epilogue"""
        if self.fusable(training):
            return F.mlp_gelu(inputs, self.fc1.kernel, self.fc1.bias, self.fc2.kernel, self.fc2.bias, residual=residual,
                              drop_path_mask=drop_path_mask)     # one tape node
        assert residual is None and drop_path_mask is None
        x = self.fc1(inputs)                     # Dense + exact-erf GELU in the GEMM epilogue
        x = self.dropout(x, training=training)
        x = self.fc2(x)
        return self.dropout(x, training=training)


class WindowAttention(Layer):
    def __init__(self, filters, window_size, num_heads, use_qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0, name=None):
        super().__init__(name=name)
        self.filters, self.window_size, self.num_heads = filters, tuple(window_size), num_heads
        self.scale = qk_scale or (filters // num_heads) ** -0.5
        self.use_qkv_bias, self.attn_drop, self.proj_drop = use_qkv_bias, attn_drop, proj_drop

    def build(self, input_shape):
        ws = self.window_size
        self.relative_position_bias_table = self.add_weight("relative_position_bias_table",
                                                            ((2 * ws[0] - 1) * (2 * ws[1] - 1), self.num_heads), "zeros")
        self.relative_position_index = _dev(relative_position_index(ws).reshape(-1))
        self.qkv = Dense(self.filters * 3, use_bias=self.use_qkv_bias, name=f"{self.name}/qkv")
        self.project = Dense(self.filters, name=f"{self.name}/proj")
        self.project_dropout = Dropout(self.proj_drop, name=f"{self.name}/project_dropout")
        self.built = True

    def call(self, x, attention_mask=None, training=None):
        C = self.filters
        qkv = self.qkv(x)                                           # [B_, N, 3C], columns [3][heads][C/heads]
        windows = 1 if attention_mask is None else attention_mask.shape[0]
        x = F.attention_packed(qkv, self.num_heads, C, C, self.scale, bias_table=self.relative_position_bias_table,
                               bias_index=self.relative_position_index, mask=attention_mask, windows=windows,
                               dropout_rate=self.attn_drop, training=bool(training),
                               bias_window=self.window_size[0] if self.window_size[0] == self.window_size[1] else 0)
        x = self.project(x)
        return self.project_dropout(x, training=training)


class SwinTransformerBlock(Layer):
    def __init__(self, filters, num_heads, window_size=7, shift_size=0, mlp_ratio=4.0, use_qkv_bias=True, qk_scale=None,
                 dropout_rate=0.0, attention_dropout_rate=0.0, drop_path_prob=0.0, norm_layer=LayerNormalization, name=None):
        super().__init__(name=name)
        self.dim, self.num_heads, self.window_size, self.shift_size, self.mlp_ratio = filters, num_heads, window_size, shift_size, mlp_ratio
        assert 0 <= self.shift_size < self.window_size, "shift_size must in 0-window_size"
        self.norm1 = norm_layer(epsilon=1e-5, name=f"{name}/norm1")
        self.attention = WindowAttention(filters, window_size=(window_size, window_size), num_heads=num_heads,
                                         use_qkv_bias=use_qkv_bias, qk_scale=qk_scale, attn_drop=attention_dropout_rate,
                                         proj_drop=dropout_rate, name=f"{name}/attn")
        self.drop_path_prob = float(drop_path_prob) if drop_path_prob > 0.0 else 0.0
        self.drop_path_masks = None      # parity tests may inject the two per-sample factor vectors
        self.norm2 = norm_layer(epsilon=1e-5, name=f"{name}/norm2")
        self.mlp = Mlp(in_features=filters, hidden_features=int(filters * mlp_ratio), dropout_rate=dropout_rate, name=f"{name}/mlp")

    def get_pad_values(self, h, w):
        return (self.window_size - h % self.window_size) % self.window_size, (self.window_size - w % self.window_size) % self.window_size

    def call(self, inputs, attention_mask=None, training=None):
        n, h, w, c = inputs.shape
        ws = self.window_size
        masks = self.drop_path_masks or (None, None)
        part, rev, hp, wp = window_index_tables(n, h, w, ws, self.shift_size)
        n_win = n * (hp // ws) * (wp // ws)
        if self.norm1.built and c % 8 == 0 and not nn.dry_run() and self.norm1.gamma.requires_grad and self.norm1.beta.requires_grad:
            # norm1 + pad + roll + partition in one pass; reverse + roll back + crop + drop path + skip connection in another; the skip
            # connection's gradient rides the LayerNorm backward (F._LnGatherFn) -- five passes over the token rows less each way
            mask = None
            if training and self.drop_path_prob != 0.0:
                mask = masks[0] if masks[0] is not None else F.drop_path_factors(n, 1.0 - self.drop_path_prob, inputs.device)
            tokens = inputs.reshape(n, h * w, c)
            link = F.residual_branch_link()
            x_windows = F.layer_norm_permute_rows(tokens, self.norm1.gamma, self.norm1.beta, self.norm1.epsilon, part, rev, (n_win, ws * ws, c), link)
            attn_windows = self.attention(x_windows, attention_mask=attention_mask if self.shift_size > 0 else None, training=training)
            x = F.permute_rows_residual(attn_windows, tokens, rev, part, mask, link)
        else:
            inputs, shortcut = F.fork(inputs, 2)      # residual fork (gradients summed by our own kernel)
            shortcut = shortcut.reshape(n, h * w, c)
            x = self.norm1(inputs)
            x_windows = F.permute_rows(x, part, rev, (n_win, ws * ws, c))      # pad + roll + partition
            attn_windows = self.attention(x_windows, attention_mask=attention_mask if self.shift_size > 0 else None, training=training)
            x = F.permute_rows(attn_windows, rev, part, (n, h * w, c))         # reverse + roll back + crop
            x = F.add(shortcut, F.drop_path(x, self.drop_path_prob, training, mask=masks[0]))
        if self.mlp.fusable(training):      # skip + drop_path(mlp(.)) out of the second product's epilogue
            mask = None
            if training and self.drop_path_prob != 0.0:
                mask = masks[1] if masks[1] is not None else F.drop_path_factors(n, 1.0 - self.drop_path_prob, x.device)
            fc1, fc2 = self.mlp.fc1, self.mlp.fc2
            params = (self.norm2.gamma, self.norm2.beta, fc1.kernel, fc1.bias, fc2.kernel, fc2.bias)
            if not nn.dry_run() and F.ln_mlp_residual_supported(x, params, mask):
                # stages of 96 / 192 channels: norm2 + MLP + drop path + skip as ONE node on the fused ConvNeXt-MLP kernels
                x = F.ln_mlp_residual(x, self.norm2.gamma, self.norm2.beta, self.norm2.epsilon, fc1.kernel, fc1.bias, fc2.kernel, fc2.bias, mask)
            else:
                x, skip = F.fork(x, 2)
                x = self.mlp(self.norm2(x), training=training, residual=skip, drop_path_mask=mask)
        else:
            x, skip = F.fork(x, 2)
            y = self.mlp(self.norm2(x), training=training)
            x = F.add(skip, F.drop_path(y, self.drop_path_prob, training, mask=masks[1]))
        return x.reshape(n, h, w, c)


class PatchMerging(Layer):
    def __init__(self, filters, norm_layer=LayerNormalization, name=None):
        super().__init__(name=name)
        self.filters = filters
        self.reduction = Dense(2 * filters, use_bias=False, name=f"{name}/reduction")
        self.norm = norm_layer(epsilon=1e-5, name=f"{name}/norm")

    def call(self, inputs, training=None):
        n, h, w, c = inputs.shape
        fwd, bwd, h2, w2 = merge_index_tables(n, h, w)
        x = F.permute_rows(inputs, fwd, bwd, (n * h2 * w2 * 4, c)).reshape(n, h2 * w2, 4 * c)
        x = self.reduction(self.norm(x))
        return x.reshape(n, h2, w2, x.shape[-1])


class BasicLayer(Layer):
    def __init__(self, filters, depth, num_heads, window_size, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path_prob=0.0, norm_layer=LayerNormalization, downsample=None, name=None):
        super().__init__(name=name)
        self.filters, self.depth = filters, depth
        self.shift_size, self.window_size = window_size // 2, window_size
        self.blocks = torch.nn.ModuleList([
            SwinTransformerBlock(filters=filters, num_heads=num_heads, window_size=window_size,
                                 shift_size=0 if (i % 2 == 0) else self.shift_size, mlp_ratio=mlp_ratio, use_qkv_bias=qkv_bias,
                                 qk_scale=qk_scale, dropout_rate=drop, attention_dropout_rate=attn_drop,
                                 drop_path_prob=drop_path_prob[i] if isinstance(drop_path_prob, list) else drop_path_prob,
                                 norm_layer=norm_layer, name=f"{name}/blocks/{i}") for i in range(depth)])
        self.downsample = downsample(filters=filters, norm_layer=norm_layer, name=f"{name}/downsample") if downsample is not None else None
        self._mask_cache = {}

    def generate_attention_mask(self, input_height, input_width, window_size, shift_size):
        key = (input_height, input_width, window_size, shift_size)
        if key not in self._mask_cache:
            self._mask_cache[key] = _dev(shift_attention_mask(input_height, input_width, window_size, shift_size))
        return self._mask_cache[key]

    def call(self, inputs, training=None):
        x = inputs
        attn_mask = None
        if not nn.dry_run():
            attn_mask = self.generate_attention_mask(inputs.shape[1], inputs.shape[2], self.window_size, self.shift_size)
        for block in self.blocks:
            x = block(x, attention_mask=attn_mask, training=training)
        before_downsample = x
        if self.downsample is not None:
            # two consumers (the endpoint list and the patch merging): forked, so their gradients are summed by our own kernel, not by the engine's add
            x, before_downsample = F.fork(x, 2)
            x = self.downsample(x, training=training)
        return x, before_downsample


class PatchEmbed(Layer):
    def __init__(self, patch_size=(4, 4), in_channels=3, embed_filters=96, norm_layer=None, name=None):
        super().__init__(name=name)
        self.patch_size, self.in_chans, self.embed_dim = tuple(patch_size), in_channels, embed_filters
        self.proj = Conv2D(embed_filters, kernel_size=patch_size, strides=patch_size, name=f"{name}/proj")
        self.norm = norm_layer(epsilon=1e-5, name=f"{name}/norm") if norm_layer is not None else None

    def call(self, x, training=None):
        h, w = x.shape[1], x.shape[2]
        pad_h = 0 if h % self.patch_size[0] == 0 else self.patch_size[0] - h % self.patch_size[0]
        pad_w = 0 if w % self.patch_size[1] == 0 else self.patch_size[1] - w % self.patch_size[1]
        p = self.proj
        if not p.built:
            p.build(tuple(x.shape))
            p.built = True
        # tf.pad (bottom / right) + "valid" conv: the padding is folded into the patch gather
        x = F.conv2d(x, p.kernel, p.bias, tuple(p.strides), (1, 1), ((0, pad_h), (0, pad_w)))
        if self.norm is not None:
            x = self.norm(x)
        return x


class SwinTransformerModel(Layer):
    def __init__(self, patch_size=(4, 4), embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4.0,
                 qkv_bias=True, qk_scale=None, dropout_rate=0.0, attention_dropout_rate=0.0, drop_path_rate=0.1,
                 norm_layer=LayerNormalization, use_absolute_pos_embed=False, patch_norm=True, return_endpoints=False,
                 name="swin_tiny_patch4_window7_224", **kwargs):
        super().__init__(name=name)
        self.patch_size, self.depths, self.num_layers, self.embed_dim = patch_size, list(depths), len(depths), embed_dim
        self.use_absolute_pos_embed, self.patch_norm = use_absolute_pos_embed, patch_norm
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.num_heads, self.window_size, self.mlp_ratio = list(num_heads), window_size, mlp_ratio
        self.qkv_bias, self.qk_scale = qkv_bias, qk_scale
        self.dropout_rate, self.attention_dropout_rate, self.drop_path_rate = dropout_rate, attention_dropout_rate, drop_path_rate
        self.norm_layer = norm_layer
        self.return_endpoints = return_endpoints

    def build(self, input_shape):
        channels = int(input_shape[-1])
        self.patch_embed = PatchEmbed(patch_size=self.patch_size, in_channels=channels, embed_filters=self.embed_dim,
                                      norm_layer=self.norm_layer if self.patch_norm else None, name="patch_embed")
        if self.use_absolute_pos_embed:      # (:563-569) zeros, one vector per patch of the BUILD resolution
            ph, pw = self.patch_size
            self._ape_grid = (-(-int(input_shape[1]) // ph), -(-int(input_shape[2]) // pw))
            self.absolute_pos_embed = self.add_weight("absolute_pos_embed", (1, self._ape_grid[0] * self._ape_grid[1], self.embed_dim), "zeros")
        self.pos_drop = Dropout(self.dropout_rate, name="postional_dropout")
        dpr = [float(x) for x in np.linspace(0.0, self.drop_path_rate, sum(self.depths))]
        layers = []
        for i in range(self.num_layers):
            layers.append(BasicLayer(filters=int(self.embed_dim * 2 ** i), depth=self.depths[i], num_heads=self.num_heads[i],
                                     window_size=self.window_size, mlp_ratio=self.mlp_ratio, qkv_bias=self.qkv_bias,
                                     qk_scale=self.qk_scale, drop=self.dropout_rate, attn_drop=self.attention_dropout_rate,
                                     drop_path_prob=dpr[sum(self.depths[:i]):sum(self.depths[:i + 1])], norm_layer=self.norm_layer,
                                     downsample=PatchMerging if (i < self.num_layers - 1) else None, name=f"layers/{i}"))
        self.basic_layers = torch.nn.ModuleList(layers)
        self.built = True

    def call(self, inputs, training=None):
        x = F.cast_input(inputs)
        x = self.patch_embed(x)
        if self.use_absolute_pos_embed:      # (:606-607) x + reshape(absolute_pos_embed, shape(x)): only the build resolution fits, as in the reference
            if tuple(x.shape[1:3]) != self._ape_grid:
                raise ValueError(f"absolute_pos_embed was built for {self._ape_grid} patches, the input gives {tuple(x.shape[1:3])}")
            x = F.add_batch_broadcast(x, self.absolute_pos_embed.reshape(1, x.shape[1], x.shape[2], x.shape[3]))
        endpoints = [x]
        x = self.pos_drop(x, training=training)
        for layer in self.basic_layers:
            x, before_downsample = layer(x, training=training)
            endpoints.append(before_downsample)
        assert len(endpoints) == len(self.basic_layers) + 1
        return endpoints if self.return_endpoints else x


def swin_tiny_224(return_endpoints=False):
    return SwinTransformerModel(name="swin_tiny_224", window_size=7, embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24],
                                drop_path_rate=0.1, return_endpoints=return_endpoints)


def swin_base_384(return_endpoints=False):
    return SwinTransformerModel(name="swin_base_384", window_size=12, embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32],
                                drop_path_rate=0.2, return_endpoints=return_endpoints)


def swin_large_384(return_endpoints=False):
    return SwinTransformerModel(name="swin_large_384", window_size=12, embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48],
                                drop_path_rate=0.3, return_endpoints=return_endpoints)
