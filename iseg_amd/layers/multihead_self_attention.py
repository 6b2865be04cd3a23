"""layers/multihead_self_attention.py of the reference (:16-203): 1x1 query / key / value projections, per-head
softmax(q k^T / sqrt(d)) with the probability clip [1e-7, 1 - 1e-7] (:138), context, reshape back to [N,H,W,C].

The reference's replace_inf on the score matrix and its NaN scrub of the probabilities are identities on finite values; here the
scores never leave the fp32 accumulators before the softmax, so those two passes are not materialised (documented deviation
for non-finite inputs only)."""
import math

from .. import functional as F
from ..nn import Layer
from .base_layers import Conv2D, Dense

EPSILON = 1e-7   # keras.backend.epsilon()


class MultiHeadSelfAttentionLayer(Layer):
    def __init__(self, filters=-1, num_heads=4, apply_linear=True, apply_scale=True, shared_qk_weights=True, shared_qk=False,
                 trainable=True, use_dense_for_linear=False, dropout_rate=0.0, use_jit_compile=False, return_attention_map=False,
                 name=None):
        super().__init__(trainable=trainable, name=name)
        self.filters, self.num_heads = filters, num_heads
        self.apply_linear, self.apply_scale = apply_linear, apply_scale
        self.shared_qk_weights, self.shared_qk = shared_qk_weights, shared_qk
        self.use_dense_for_linear = use_dense_for_linear
        self.dropout_rate = dropout_rate
        self.return_attention_map = bool(return_attention_map)

    def build(self, input_shape):
        channels = int(input_shape[-1])
        qk_filters = channels if self.filters == -1 else self.filters
        self.qk_filters, self.channels = qk_filters, channels

        def linear(units, name, init="glorot_uniform"):
            if self.use_dense_for_linear:
                return Dense(units, kernel_initializer=init, trainable=self.trainable, name=f"{self.name}/{name}")
            return Conv2D(units, (1, 1), kernel_initializer=init, trainable=self.trainable, name=f"{self.name}/{name}")

        if self.apply_linear:
            self.query_conv = linear(qk_filters, "query_conv", ("uniform", 0.05))
            if not self.shared_qk:
                self.key_conv = linear(qk_filters, "key_conv", ("uniform", 0.05))
            self.value_conv = linear(channels, "value_conv")
            for layer in (self.query_conv, getattr(self, "key_conv", None), self.value_conv):
                if layer is not None:
                    layer.build(input_shape)
                    layer.built = True
            if self.shared_qk_weights and not self.shared_qk:   # SharedInitializer: same initial values (:66-73)
                self.key_conv.kernel.data.copy_(self.query_conv.kernel.data)
        self.built = True

    def compute_attention(self, query, key, value, attention_mask=None, training=None):
        """(:108-151) attention_mask (1 = attend, 0 = masked; [HW, HW], [N, HW, HW] or [N, 1, HW, HW]) enters safed_softmax as
        (1 - mask) * -1e9 (utils/op_utils.py:24-38); with return_attention_map the probabilities that multiply V -- after dropout and the
        [1e-7, 1 - 1e-7] clip -- come back as a second result [N, heads, HW, HW] (no gradient flows through it)"""
        import torch

        n, h, w, _ = query.shape
        cq, cv = query.shape[-1], value.shape[-1]
        query = F.replace_nan_or_inf(query, EPSILON)
        key = F.replace_nan_or_inf(key, EPSILON)
        qkv = F.concat([query, key, value]).reshape(n, h * w, 2 * cq + cv)
        scale = 1.0 / math.sqrt(cq // self.num_heads) if self.apply_scale else 1.0
        mask, windows = None, 1
        if attention_mask is not None:
            m = attention_mask.to(device=qkv.device, dtype=torch.float32)
            if m.dim() == 4:
                if m.shape[1] != 1:
                    raise NotImplementedError("attention_mask per head: [N, 1, HW, HW], [N, HW, HW] or [HW, HW] are supported")
                m = m[:, 0]
            if m.dim() == 2:
                m = m[None]
            if tuple(m.shape[1:]) != (h * w, h * w) or m.shape[0] not in (1, n):
                raise ValueError(f"attention_mask of shape {tuple(attention_mask.shape)} does not fit {n} x {h * w} x {h * w} scores")
            mask, windows = ((1.0 - m) * -1e9).contiguous(), int(m.shape[0])
        out = F.attention_packed(qkv, self.num_heads, cq, cv, scale, mask=mask, windows=windows, clip=(EPSILON, 1.0 - EPSILON),
                                 dropout_rate=self.dropout_rate, training=bool(training), return_probs=self.return_attention_map)
        x, probs = out if self.return_attention_map else (out, None)
        x = F.replace_nan_or_inf(x.reshape(n, h, w, cv), EPSILON)
        return (x, probs) if self.return_attention_map else x

    def call(self, inputs, key=None, value=None, attention_mask=None, training=None):
        # (:153-203) the reference's call() accepts attention_mask but never hands it to compute_attention; neither does this one --
        # compute_attention(..., attention_mask=...) is the entry that honours it
        query = inputs
        if key is None:
            key = query
        if value is None:
            value = key
        if self.apply_linear:
            query = self.query_conv(query)
            key = query if self.shared_qk else self.key_conv(key)
            value = self.value_conv(value)
        return self.compute_attention(query, key, value, training=training)
