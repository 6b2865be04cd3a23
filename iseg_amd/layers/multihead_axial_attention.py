"""layers/multihead_axial_attention.py of the reference (:15-172): 1x1 query / key / value projections; per head a column attention map
softmax(q k^T / sqrt(d)) over the H positions of every column and a row map over the W positions of every row, both from the SAME
projected query / key, both clipped to [1e-7, 1 - 1e-7]; the value is mixed along H first, then along W (:127-132), and the heads are
interleaved channel-minor on the way out (:137-139: [N, H, W, C/heads, heads] -> channel index c * heads + head).

Built on the batched attention node of the other attention layers (functional.attention_packed: strided-batch MFMA GEMMs for the scores
and the context, row softmax + clip kernels) -- the column pass sees the map as N*W sequences of H tokens through a static row permutation
(functional.permute_rows, the Swin window-partition kernel), the row pass as N*H sequences of W tokens with no data movement; the head
interleave is a product with a fixed 0/1 matrix (exact in every storage type).  As in the self-attention layer, the reference's
replace_inf / replace_nan passes over the scores and probabilities are identities on finite values and are not materialised."""
import math

import torch

from .. import functional as F
from .. import kernels as K
from .. import nn as _nn
from ..nn import Layer
from .base_layers import Conv2D

EPSILON = 1e-7   # keras.backend.epsilon()


class _ChannelPermuteFn(torch.autograd.Function):
    """y = x P with a fixed 0/1 matrix P [C, C] in the activation type (one MFMA product each way: exact, every output is one input)"""

    @staticmethod
    def forward(ctx, x, P):
        C = x.shape[-1]
        x2 = x.contiguous().reshape(-1, C)
        y = torch.empty_like(x2)
        K.gemm(x2, P, y, x2.shape[0], C, C, lda=C, ldb=C, ldd=C, a_kcontig=1, b_kcontig=0)
        ctx.P = P
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        C = dy.shape[-1]
        d2 = dy.contiguous().reshape(-1, C)
        dx = torch.empty_like(d2)
        K.gemm(d2, ctx.P, dx, d2.shape[0], C, C, lda=C, ldb=C, ldd=C, a_kcontig=1, b_kcontig=1)      # dy P^T: P's rows are the reduction
        return dx.reshape(dy.shape), None


class MultiHeadAxialAttentionLayer(Layer):
    def __init__(self, filters=-1, num_heads=4, apply_linear=True, apply_scale=True, shared_qk_weights=True, shared_qk=False, trainable=True,
                 linear_func=None, name=None):
        super().__init__(trainable=trainable, name=name)
        self.filters, self.num_heads = filters, num_heads
        self.apply_linear, self.apply_scale = apply_linear, apply_scale
        self.shared_qk_weights, self.shared_qk = shared_qk_weights, shared_qk
        self.linear_func = linear_func or Conv2D      # (the reference's default is keras.layers.Conv2D)
        self._tables = {}

    def build(self, input_shape):
        channels = int(input_shape[-1])
        qk_filters = channels if self.filters == -1 else self.filters
        if qk_filters % self.num_heads or channels % self.num_heads:
            raise ValueError(f"{self.name}: {qk_filters} query / {channels} value channels do not split into {self.num_heads} heads")
        self.qk_filters, self.channels = qk_filters, channels
        if self.apply_linear:
            self.query_conv = self.linear_func(qk_filters, (1, 1), trainable=self.trainable, name=f"{self.name}/query_conv")
            if not self.shared_qk:
                self.key_conv = self.linear_func(qk_filters, (1, 1), trainable=self.trainable, name=f"{self.name}/key_conv")
            self.value_conv = self.linear_func(channels, (1, 1), trainable=self.trainable, name=f"{self.name}/value_conv")
            for layer in (self.query_conv, getattr(self, "key_conv", None), self.value_conv):
                if layer is not None:
                    layer.build(input_shape)
                    layer.built = True
            if self.shared_qk_weights and not self.shared_qk:   # SharedInitializer: same initial values (:50-57)
                self.key_conv.kernel.data.copy_(self.query_conv.kernel.data)
        self.built = True

    def _transpose_tables(self, n, h, w, device):
        """row tables of [N, H, W] -> [N, W, H] and back (int32, cached per shape and device)"""
        key = (n, h, w, str(device))
        if key not in self._tables:
            idx = torch.arange(n * h * w, dtype=torch.int32).reshape(n, h, w)
            fwd = idx.permute(0, 2, 1).reshape(-1).contiguous().to(device)               # destination (n, w, h) <- source (n, h, w)
            bwd = torch.arange(n * w * h, dtype=torch.int32).reshape(n, w, h).permute(0, 2, 1).reshape(-1).contiguous().to(device)
            self._tables[key] = (fwd, bwd)
        return self._tables[key]

    def _interleave(self, cv, like):
        """[cv, cv] 0/1 matrix that moves input channel head * d + c to output channel c * heads + head (:137-139)"""
        key = ("interleave", cv, like.dtype, str(like.device))
        if key not in self._tables:
            d = cv // self.num_heads
            src = torch.arange(cv)
            dst = (src % d) * self.num_heads + src // d
            m = torch.zeros(cv, cv, dtype=torch.float32)
            m[src, dst] = 1.0
            self._tables[key] = m.to(device=like.device, dtype=like.dtype)
        return self._tables[key]

    def compute_attetnion(self, query, key, value, training=None):      # (sic: the reference's spelling, :84)
        n, h, w, cv = value.shape
        cq = query.shape[-1]
        query = F.replace_nan_or_inf(query, EPSILON)
        key = F.replace_nan_or_inf(key, EPSILON)
        scale = 1.0 / math.sqrt(cq // self.num_heads) if self.apply_scale else 1.0
        clip = (EPSILON, 1.0 - EPSILON)
        # both passes read the same projected query / key
        if key is query:
            q0, k0, q1, k1 = F.fork(query, 4)
        else:
            (q0, q1), (k0, k1) = F.fork(query, 2), F.fork(key, 2)
        # column pass: every (sample, column) is a sequence of H tokens
        fwd, bwd = (None, None) if _nn.dry_run() else self._transpose_tables(n, h, w, value.device)
        cols = F.permute_rows(F.concat([q0, k0, value]), fwd, bwd, (n * w, h, 2 * cq + cv))
        x = F.attention_packed(cols, self.num_heads, cq, cv, scale, clip=clip)
        x = F.permute_rows(x, bwd, fwd, (n, h, w, cv))
        # row pass on the column pass's result: every (sample, row) is a sequence of W tokens
        rows = F.concat([q1, k1, x]).reshape(n * h, w, 2 * cq + cv)
        x = F.attention_packed(rows, self.num_heads, cq, cv, scale, clip=clip).reshape(n, h, w, cv)
        if self.num_heads > 1 and not _nn.dry_run():
            x = _ChannelPermuteFn.apply(x, self._interleave(cv, x))
        return F.replace_nan_or_inf(x, EPSILON)

    compute_attention = compute_attetnion

    def call(self, inputs, training=None):
        x = inputs
        if self.apply_linear:
            query = self.query_conv(x)
            key = query if self.shared_qk else self.key_conv(x)
            x = self.value_conv(x)
        else:
            query = key = x
        return self.compute_attetnion(query, key, x, training=training)
