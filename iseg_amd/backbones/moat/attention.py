"""backbones/moat/attention.py of the reference: TrailDense (:104-187) and Attention (:190-339) -- multi-head self-attention over all tokens of a
window with head projections kept as [C, heads, head_size] / [heads, head_size, C] kernels ("weight") and biases, q scaled by head_size^-0.5, softmax
in fp32.

The 2-D relative position embedding (:229-277; `use_pos_emb=True`, the moatN constructors' default, get_backbone passes False): a trainable table
[heads, 2 P - 1, 2 P - 1] (P = position_embedding_size: 14 at stride 16, 7 at stride 32) that build() resizes bilinearly to [2 h - 1, 2 w - 1]
(tf.image.resize, half-pixel centres) and re-indexes to one additive bias per head, bias[n, (i, j), (x, y)] = R[n, x - i + h - 1, y - j + w - 1]
(reindex_2d_einsum_lookup :68-120).  The reference computes that bias ONCE, in build(), from the table's value at that moment -- Keras runs build()
in an init scope, so `self.reindexed_position_embedding` is a constant: no gradient reaches the table and later updates of the table (weight decay,
a checkpoint loaded after the first call) do not reach the logits.  Restated as such: the bias is derived on the host from the table's current value
at the first real call (a [T T, heads] fp32 look-up with an identity index for the attention node's bias gather), the table parameter stays in the
model (and in checkpoints) and receives no gradient; `reset_position_bias()` re-derives it after a weight import."""
from ... import functional as F
from ...nn import Layer

_INIT = ("truncated_normal", 0.02)      # tf.random_normal_initializer(stddev=0.02): parity tests inject weights, training starts from N(0, 0.02)


def resize_bilinear_half_pixel(table, out_h, out_w):
    """tf.image.resize(method="bilinear") of a [n, h, w] float32 array (half-pixel centres, no antialias): host arithmetic for a build-time table"""
    import numpy as np

    n, h, w = table.shape

    def taps(size_in, size_out):
        src = (np.arange(size_out, dtype=np.float32) + np.float32(0.5)) * np.float32(size_in / size_out) - np.float32(0.5)
        lo = np.floor(src)
        frac = (src - lo).astype(np.float32)
        i0 = np.clip(lo, 0, size_in - 1).astype(np.int64)
        i1 = np.clip(lo + 1, 0, size_in - 1).astype(np.int64)
        return i0, i1, frac

    y0, y1, fy = taps(h, out_h)
    x0, x1, fx = taps(w, out_w)
    top = table[:, y0][:, :, x0] * (1 - fx) + table[:, y0][:, :, x1] * fx
    bot = table[:, y1][:, :, x0] * (1 - fx) + table[:, y1][:, :, x1] * fx
    return (top * (1 - fy)[None, :, None] + bot * fy[None, :, None]).astype(np.float32)


def reindex_2d(table, height, width):
    """reindex_2d_einsum_lookup (:68-120) with max_relative_height / width = height - 1 / width - 1: [n, 2 h - 1, 2 w - 1] -> [n, h w, h w]"""
    import numpy as np

    ih = np.arange(height)[None, :] - np.arange(height)[:, None] + height - 1      # [i, x]
    iw = np.arange(width)[None, :] - np.arange(width)[:, None] + width - 1         # [j, y]
    out = table[:, ih[:, None, :, None], iw[None, :, None, :]]                      # [n, i, j, x, y]
    return out.reshape(table.shape[0], height * width, height * width)


class Attention(Layer):
    def __init__(self, hidden_size, head_size, relative_position_embedding_type=None, scale_ratio=None, name="attention", trainable=True, **kwargs):
        super().__init__(name=name, trainable=trainable)
        if relative_position_embedding_type not in (None, "2d_multi_head"):
            raise ValueError(f"MOAT Attention: relative_position_embedding_type {relative_position_embedding_type!r} (None or '2d_multi_head', :264-277)")
        self.relative_position_embedding_type, self.scale_ratio = relative_position_embedding_type, scale_ratio
        self.hidden_size, self.head_size = int(hidden_size), int(head_size)
        self.num_heads = self.hidden_size // self.head_size
        self._q_scale = self.head_size ** -0.5
        self.relative_position_embedding = None
        self._pos_bias = None      # (bias look-up [T T, heads] fp32, identity index [T T] int32, (h, w)) once derived

    def build(self, input_shape):
        c, h, d = int(input_shape[-1]), self.num_heads, self.head_size
        if self.relative_position_embedding_type == "2d_multi_head":
            if len(input_shape) != 4:
                raise ValueError("The input shape should be [batch_size, height, width, channels]")
            height, width = int(input_shape[-3]), int(input_shape[-2])
            if self.scale_ratio is not None:
                if isinstance(self.scale_ratio, (list, tuple)) and len(self.scale_ratio) == 2:
                    hs, ws = self.scale_ratio
                elif isinstance(self.scale_ratio, float):
                    hs = ws = self.scale_ratio
                else:
                    raise ValueError("scale ratio should be float or list of floats with length 2")
                eh, ew = 2 * round(height / hs) - 1, 2 * round(width / ws) - 1
            else:
                eh, ew = 2 * height - 1, 2 * width - 1
            self._pos_hw = (height, width)
            self.relative_position_embedding = self.add_weight("relative_position_embedding", (h, eh, ew), _INIT)
        self.q_weight = self.add_weight("q/weight", (c, h, d), _INIT)
        self.q_bias = self.add_weight("q/bias", (h, d), "zeros")
        self.k_weight = self.add_weight("k/weight", (c, h, d), _INIT)
        self.k_bias = self.add_weight("k/bias", (h, d), "zeros")
        self.v_weight = self.add_weight("v/weight", (c, h, d), _INIT)
        self.v_bias = self.add_weight("v/bias", (h, d), "zeros")
        self.o_weight = self.add_weight("o/weight", (h, d, self.hidden_size), _INIT)
        self.o_bias = self.add_weight("o/bias", (self.hidden_size,), "zeros")
        self.built = True

    def reset_position_bias(self):
        """derive the position bias again from the table's current value at the next call (after a weight import)"""
        self._pos_bias = None

    def _position_bias(self, height, width, device):
        import numpy as np
        import torch

        if (height, width) != self._pos_hw:
            raise ValueError(f"MOAT Attention: built for {self._pos_hw[0]} x {self._pos_hw[1]} tokens, called with {height} x {width} (the reference's "
                             "re-indexed position embedding is a build-time constant of the first input size, :258-306)")
        if self._pos_bias is None:
            table = self.relative_position_embedding.data.detach().float().cpu().numpy()
            if self.scale_ratio is not None:
                table = resize_bilinear_half_pixel(table, 2 * height - 1, 2 * width - 1)
            bias = reindex_2d(table, height, width)                                             # [heads, T, T]
            T = height * width
            look = torch.from_numpy(np.ascontiguousarray(bias.reshape(self.num_heads, T * T).T)).to(device)      # [T T, heads]: bias[h, i, j] = look[i T + j, h]
            index = torch.arange(T * T, dtype=torch.int32, device=device)
            self._pos_bias = (look, index)
        return self._pos_bias

    def call(self, query, training=None):
        from ... import nn

        b, hh, ww, c = query.shape
        h, d = self.num_heads, self.head_size
        x = query.reshape(b, hh * ww, c)
        qkv = F.dense_group(x, [self.q_weight, self.k_weight, self.v_weight], [self.q_bias, self.k_bias, self.v_bias])
        if self.relative_position_embedding is not None and not nn.dry_run():
            look, index = self._position_bias(hh, ww, query.device)
            y = F.attention_packed(qkv, h, h * d, h * d, self._q_scale, bias_table=look, bias_index=index)
        else:
            y = F.attention_packed(qkv, h, h * d, h * d, self._q_scale)
        return F.dense(y, self.o_weight, self.o_bias, kshape=(h * d, self.hidden_size))
