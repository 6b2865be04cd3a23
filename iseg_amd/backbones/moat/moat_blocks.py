"""backbones/moat/moat_blocks.py of the reference: drop_connect (:9-34), residual_add_with_drop_path (:36-45), SqueezeAndExcitation (:48-95),
MBConvBlock (:98-245), MOATBlock (:248-508)."""
import math

import torch

from ... import functional as F
from ... import nn as _nn
from ...layers.base_layers import Conv2D, DepthwiseConv2D, LayerNormalization
from ...layers.model_builder import get_training_value
from ...layers.nasfpn import _ChannelGateFn, _SigmoidGateFn
from ...layers.normalizations import normalization
from ...nn import Layer
from .attention import Attention

_INIT = ("truncated_normal", 0.02)


def residual_add_with_drop_path(residual, shortcut, survival_prob, training, mask=None):
    """(:36-45) drop_connect = x / survival_prob * floor(survival_prob + U[0, 1)) per sample = this package's drop path with keep = survival_prob"""
    if survival_prob is not None and 0 < survival_prob < 1:
        residual = F.drop_path(residual, 1.0 - survival_prob, training, mask=mask)
    return F.add(shortcut, residual)


class SqueezeAndExcitation(Layer):
    def __init__(self, se_filters, output_filters, activation="swish", name="se", trainable=True):
        super().__init__(name=name, trainable=trainable)
        self._se_reduce = Conv2D(se_filters, 1, padding="same", use_bias=True, kernel_initializer=_INIT, name=f"{self.name}/reduce_conv2d")
        self._se_expand = Conv2D(output_filters, 1, padding="same", use_bias=True, kernel_initializer=_INIT, name=f"{self.name}/expand_conv2d")
        self.activation = activation

    def call(self, inputs, training=None):
        x, gated = F.fork(inputs, 2)
        g = self._se_reduce(F.global_avg_pool(x))
        g = F.swish(g) if self.activation == "swish" else F.gelu(g)
        g = self._se_expand(g)
        if _nn.dry_run():
            return gated
        return _ChannelGateFn.apply(gated, _SigmoidGateFn.apply(g))


class _MBConvPart(Layer):
    """the mobile-convolution half both block types share: shortcut (average pool + 1 x 1 projection), pre-norm -> 1 x 1 expand -> norm -> gelu ->
    depthwise k x k (stride) -> norm -> gelu -> [squeeze-and-excitation] -> 1 x 1 shrink"""

    def _build_mbconv(self, input_size, with_se):
        inner = self.hidden_size * self.expansion_rate
        n = self.name
        self._shortcut_conv = (Conv2D(self.hidden_size, 1, padding="same", use_bias=True, kernel_initializer=_INIT, name=f"{n}/shortcut_conv")
                               if input_size != self.hidden_size else None)
        self._pre_norm = self._norm_class(name=f"{n}/pre_norm")
        self._expand_conv = Conv2D(inner, 1, padding="same", use_bias=False, kernel_initializer=_INIT, name=f"{n}/expand_conv")
        self._expand_norm = self._norm_class(name=f"{n}/expand_norm")
        self._depthwise_conv = DepthwiseConv2D(self.kernel_size, strides=self.block_stride, padding="same", use_bias=False, depthwise_initializer=_INIT,
                                               name=f"{n}/depthwise_conv")
        self._depthwise_norm = self._norm_class(name=f"{n}/depthwise_norm")
        self._se = None
        if with_se:
            self._se = SqueezeAndExcitation(max(1, int(self.hidden_size * self.se_ratio)), inner, name=f"{n}/se")
        self._shrink_conv = Conv2D(self.hidden_size, 1, padding="same", use_bias=True, kernel_initializer=_INIT, name=f"{n}/shrink_conv")

    def _shortcut_branch(self, x):
        if self.block_stride > 1:      # (:199-214: AveragePooling2D(pool_size, strides=block_stride, "same"), evaluated in fp32 by the reference)
            x = F.avg_pool2d(x, self.pool_size, strides=self.block_stride, padding="same")
        return self._shortcut_conv(x) if self._shortcut_conv is not None else x

    def _mbconv(self, inputs, training):
        a, b = F.fork(inputs, 2)
        shortcut = self._shortcut_branch(a)
        x = self._pre_norm(b, training=training)
        x = F.gelu(self._expand_norm(self._expand_conv(x), training=training))
        x = F.gelu(self._depthwise_norm(self._depthwise_conv(x), training=training))
        if self._se is not None:
            x = self._se(x)
        return self._shrink_conv(x), shortcut


class MBConvBlock(_MBConvPart):
    def __init__(self, hidden_size, kernel_size=3, expansion_rate=4, se_ratio=0.25, block_stride=1, pool_size=2, norm_class=normalization,
                 activation="gelu", survival_prob=None, name="mbconv", trainable=True, **kwargs):
        super().__init__(name=name, trainable=trainable)
        self.hidden_size, self.kernel_size, self.expansion_rate, self.se_ratio = hidden_size, kernel_size, expansion_rate, se_ratio
        self.block_stride, self.pool_size, self._norm_class, self.survival_prob = block_stride, pool_size, norm_class, survival_prob
        self.drop_path_mask = None

    def build(self, input_shape):
        self._build_mbconv(int(input_shape[-1]), self.se_ratio is not None)
        self.built = True

    def call(self, inputs, training=None):
        training = get_training_value(training)
        x, shortcut = self._mbconv(inputs, training)
        return residual_add_with_drop_path(x, shortcut, self.survival_prob, training, self.drop_path_mask)


class MOATBlock(_MBConvPart):
    def __init__(self, hidden_size, kernel_size=3, expansion_rate=4, block_stride=2, pool_size=2, norm_class=normalization, activation="gelu",
                 head_size=32, window_size=None, relative_position_embedding_type="2d_multi_head", position_embedding_size=7, ln_epsilon=1e-5,
                 survival_prob=None, use_checkpointing_for_attention=False, name="moat", trainable=True, **kwargs):
        super().__init__(name=name, trainable=trainable)
        # window_size = [height, width] (:317-327): the attention runs inside non-overlapping windows (_make_windows / _remove_windows, :407-434);
        # None: one window = the whole (strided) map, as every moat0-4 constructor leaves it (backbones/moat/moat.py:245-299)
        if window_size and not (isinstance(window_size, (list, tuple)) and len(window_size) == 2):
            raise ValueError("The window size should be a list of two ints [height, width], if specified.")
        self.window_size = [int(v) for v in window_size] if window_size else None
        self._window_tables = {}
        self.hidden_size, self.kernel_size, self.expansion_rate = hidden_size, kernel_size, expansion_rate
        self.block_stride, self.pool_size, self._norm_class = block_stride, pool_size, norm_class
        self.head_size, self.ln_epsilon, self.survival_prob = head_size, ln_epsilon, survival_prob
        self.relative_position_embedding_type, self.position_embedding_size = relative_position_embedding_type, position_embedding_size
        self.se_ratio = None
        self.drop_path_masks = None

    def build(self, input_shape):
        height, width, input_size = (int(v) for v in input_shape[-3:])
        if self.window_size:
            self._window_height, self._window_width = self.window_size
        else:
            self._window_height = math.ceil(float(height) / self.block_stride)
            self._window_width = math.ceil(float(width) / self.block_stride)
        self._build_mbconv(input_size, False)
        self._attention_norm = LayerNormalization(epsilon=self.ln_epsilon, name=f"{self.name}/attention_norm")
        if self.relative_position_embedding_type and self.position_embedding_size is None:
            raise ValueError("The position embedding size need to be specified if relative position embedding is used.")
        scale_ratio = None
        if self.relative_position_embedding_type:      # (:380-392)
            scale_ratio = [self._window_height / self.position_embedding_size, self._window_width / self.position_embedding_size]
        self._attention = Attention(hidden_size=self.hidden_size, head_size=self.head_size,
                                    relative_position_embedding_type=self.relative_position_embedding_type, scale_ratio=scale_ratio,
                                    name=f"{self.name}/attention")
        self.built = True

    def call(self, inputs, training=None):
        training = get_training_value(training)
        masks = self.drop_path_masks or (None, None)
        x, shortcut = self._mbconv(inputs, training)
        x = residual_add_with_drop_path(x, shortcut, self.survival_prob, training, masks[0])
        x, attention_shortcut = F.fork(x, 2)
        b, h, w, c = x.shape
        y = self._attention_norm(x)
        if self.window_size:
            part, rev = self._partition_tables(b, h, w, y.device)
            wh, ww = self._window_height, self._window_width
            y = F.permute_rows(y.reshape(b * h * w, c), part, rev, (b * (h // wh) * (w // ww), wh, ww, c))      # _make_windows
            y = self._attention(y, training=training)
            y = F.permute_rows(y.reshape(b * h * w, c), rev, part, (b, h, w, c))                                  # _remove_windows
        else:
            y = self._attention(y, training=training)
            y = y.reshape(b, h, w, c)
        return residual_add_with_drop_path(y, attention_shortcut, self.survival_prob, training, masks[1])

    def _partition_tables(self, b, h, w, device):
        """row tables of the window partition (:407-424): windowed row r = ((n, wy, wx), (iy, ix)) takes map row (n, wy wh + iy, wx ww + ix); the
        reverse table is its inverse (a permutation: every map pixel sits in exactly one window)"""
        key = (b, h, w, str(device))
        if key not in self._window_tables:
            wh, ww = self._window_height, self._window_width
            if h % wh or w % ww:
                raise ValueError(f"MOATBlock(window_size={self.window_size}): the {h} x {w} map is not a whole number of windows")
            idx = torch.arange(b * h * w, dtype=torch.int64).reshape(b, h // wh, wh, w // ww, ww).permute(0, 1, 3, 2, 4).reshape(-1)
            inv = torch.empty_like(idx)
            inv[idx] = torch.arange(idx.numel(), dtype=torch.int64)
            self._window_tables[key] = (idx.to(torch.int32).to(device), inv.to(torch.int32).to(device))
        return self._window_tables[key]
