"""layers/nasfpn.py of the reference (:33-406, the TF Model Garden decoder it adapts): the NAS-FPN feature pyramid.

Levels `min_level .. max_level` are seeded from the backbone endpoints (1x1 convolution + BatchNorm where the channel count differs from
`num_filters`, :248-262; missing coarse levels by a stride-2 max-pool of the level below, :218-232), then `num_repeats` cells of the seven
searched merge blocks run (NASFPN_BLOCK_SPECS, :37-45): two earlier nodes are resampled to the block's level (max-pool down, nearest
up-sampling up, :264-271), combined -- a sum, or the "global attention" of :304-311 (`feat0 + feat1 * sigmoid(max over H, W of feat0)`, the
coarser node gating the finer one) when `use_sum_for_combination` is off -- joined by every still-unused node of the same level when the
block is an output (:348-355), then activation -> 3x3 convolution -> BatchNorm (:357-371).  The last five nodes of a cell are its levels.

Everything is composed from this package's kernels: the implicit-GEMM convolution, (Sync)BatchNorm, the pooling kernels, row gathers.
  * nearest up-sampling by an integer factor is a row gather (one index table per shape); its gradient is the sum over each s x s cell =
    s^2 x the average pool with window = stride = s;
  * the maximum over H x W is a chain of max-pools with windows of at most 15 x 15 cells (padding="same" pads with -inf, so ragged sizes
    work): each link takes the pooling kernels' two-pass gradient (one winner byte per window).  A tie sends the gradient to the first
    maximal cell (TF's MaxPoolGrad rule) where `reduce_max` would split it between the tied cells -- conv + BatchNorm outputs do not tie;
  * the sigmoid of the [N, C] maxima is the C ABI's activation kernel (`iseg_act_fwd` / `iseg_act_bwd`, ISEG_ACT_SIGMOID) in fp32;
  * the per-sample channel gate is `iseg_scale_cols` per sample, its gradient `iseg_mul_colsum` per sample.
Both convolution variants are built: plain `Conv2D` and `use_separable_conv=True` (keras SeparableConv2D = depthwise k x k + pointwise 1 x 1, :176-181,
283-302), with any activation `keras.activations.get` of this package knows (relu, gelu, swish / silu, sigmoid; :194).  Initialisers follow
:283-302 (VarianceScaling(2, fan_out, untruncated normal), zero biases); weight regularisers are not modelled by this package and raise."""
import torch

from .. import functional as F
from .. import kernels as K
from .. import nn as _nn
from ..nn import Layer
from .base_layers import BatchNormalization, Conv2D, SeparableConv2D, get_activation

# (block_level, combine_fn, (input_offset0, input_offset1), is_output) -- nasfpn.py:37-45
NASFPN_BLOCK_SPECS = [
    (4, "attention", (1, 3), False),
    (4, "sum", (1, 5), False),
    (3, "sum", (0, 6), True),
    (4, "sum", (6, 7), True),
    (5, "attention", (7, 8), True),
    (7, "attention", (6, 9), True),
    (6, "attention", (9, 10), True),
]


class BlockSpec:
    def __init__(self, level, combine_fn, input_offsets, is_output):
        self.level, self.combine_fn, self.input_offsets, self.is_output = level, combine_fn, tuple(input_offsets), is_output


def build_block_specs(block_specs=None):
    return [BlockSpec(*b) for b in (block_specs or NASFPN_BLOCK_SPECS)]


_TABLES = {}


def _upsample_table(n, h, w, scale, device):
    key = (n, h, w, scale, str(device))
    if key not in _TABLES:
        src = torch.arange(n * h * w, dtype=torch.int32).reshape(n, h, 1, w, 1)
        _TABLES[key] = src.expand(n, h, scale, w, scale).reshape(-1).contiguous().to(device)
    return _TABLES[key]


class _NearestUpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        n, h, w, c = x.shape
        ctx.scale = scale
        idx = _upsample_table(n, h, w, scale, x.device)
        return K.gather_rows(x.contiguous().reshape(-1, c), idx, idx.numel()).reshape(n, h * scale, w * scale, c)

    @staticmethod
    def backward(ctx, dy):
        s = ctx.scale
        n, hs, ws, c = dy.shape
        pooled = K.pool2d_fwd(dy.contiguous(), s, s, s, s, 0, 0, hs // s, ws // s, K.POOL_AVG)
        return K.axpby(pooled, pooled, float(s * s), 0.0), None      # the sum over each s x s cell


def nearest_upsampling(data, scale):
    """[N, H, W, C] -> [N, H * scale, W * scale, C], every pixel repeated scale x scale times (nasfpn.py:48-84)"""
    if scale == 1:
        return data
    if _nn.dry_run():
        return data.new_empty((data.shape[0], data.shape[1] * scale, data.shape[2] * scale, data.shape[3]))
    return _NearestUpFn.apply(data, int(scale))


def global_max(x):
    """max over H and W, [N, H, W, C] -> [N, 1, 1, C], as a chain of max-pools of at most 15 x 15 cells"""
    while x.shape[1] > 1 or x.shape[2] > 1:
        kh, kw = min(int(x.shape[1]), 15), min(int(x.shape[2]), 15)
        x = F.max_pool2d(x, (kh, kw), strides=(kh, kw), padding="same")
    return x


class _SigmoidGateFn(torch.autograd.Function):
    """m = sigmoid(x) for the [N, 1, 1, C] maxima -> fp32 [N, C] (C-ABI kernels: cast, iseg_act_fwd / iseg_act_bwd with ISEG_ACT_SIGMOID)"""

    @staticmethod
    def forward(ctx, x):
        pre = x.reshape(x.shape[0], x.shape[-1]).contiguous()
        pre = pre if pre.dtype == torch.float32 else K.cast(pre, torch.float32)
        ctx.save_for_backward(pre)
        ctx.shape, ctx.dtype = x.shape, x.dtype
        return K.act_fwd(pre, K.ACT_SIGMOID)

    @staticmethod
    def backward(ctx, dm):
        (pre,) = ctx.saved_tensors
        dx = K.act_bwd(dm.contiguous(), pre, K.ACT_SIGMOID)
        dx = dx if ctx.dtype == torch.float32 else K.cast(dx, ctx.dtype)
        return dx.reshape(ctx.shape)


class _ChannelGateFn(torch.autograd.Function):
    """y[n, h, w, c] = x[n, h, w, c] * m[n, c]"""

    @staticmethod
    def forward(ctx, x, m):
        n, h, w, c = x.shape
        xc = x.contiguous()
        ctx.save_for_backward(xc, m)
        y = torch.empty_like(xc)
        for i in range(n):
            y[i].reshape(h * w, c).copy_(K.scale_cols(xc[i].reshape(h * w, c), m[i].contiguous()))
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, m = ctx.saved_tensors
        n, h, w, c = xc.shape
        dyc = dy.contiguous()
        dx = torch.empty_like(xc)
        dm = torch.zeros_like(m)
        for i in range(n):
            dx[i].reshape(h * w, c).copy_(K.scale_cols(dyc[i].reshape(h * w, c), m[i].contiguous()))
            K.mul_colsum(dyc[i].reshape(h * w, c), xc[i].reshape(h * w, c), dm[i], accumulate=False)
        return dx, dm


def global_attention(feat0, feat1):
    """nasfpn.py:304-311: feat0 + feat1 * sigmoid(max over H, W of feat0)"""
    if _nn.dry_run():
        return feat0
    f0a, f0b = F.fork(feat0, 2)
    m = _SigmoidGateFn.apply(global_max(f0a))
    return F.add(f0b, _ChannelGateFn.apply(feat1, m))


class NASFPN(Layer):
    """call(inputs): {str(level): [N, H / 2^level, W / 2^level, C_level]} -> {str(level): [N, ., ., num_filters]} for min_level..max_level"""

    def __init__(self, input_specs, min_level=3, max_level=7, block_specs=None, use_sum_for_combination=True, num_filters=256, num_repeats=5,
                 use_separable_conv=False, activation="relu", use_sync_bn=False, norm_momentum=0.99, norm_epsilon=0.001,
                 kernel_initializer="VarianceScaling", kernel_regularizer=None, bias_regularizer=None, name="nasfpn", trainable=True, **kwargs):
        super().__init__(name=name, trainable=trainable)
        if kernel_regularizer is not None or bias_regularizer is not None:
            raise NotImplementedError("NASFPN: kernel_regularizer / bias_regularizer are not modelled by this package (no layer of it adds a weight penalty)")
        self.use_separable_conv = bool(use_separable_conv)
        self.activation = get_activation(activation) or (lambda t: t)
        if min(str(k) for k in input_specs.keys()) > str(min_level):
            raise ValueError("Backbone min level should be less or equal to FPN min level")      # (nasfpn.py:237-239)
        self.input_specs = {str(k): tuple(v) for k, v in input_specs.items()}
        self.min_level, self.max_level = int(min_level), int(max_level)
        self.block_specs = build_block_specs() if block_specs is None else list(block_specs)
        self.use_sum_for_combination = bool(use_sum_for_combination)
        self.num_filters, self.num_repeats = int(num_filters), int(num_repeats)
        self.norm_kwargs = dict(momentum=norm_momentum, epsilon=norm_epsilon, synchronized=bool(use_sync_bn))
        self._build_layers()

    def _conv(self, filters, kernel_size, name):
        """(:176-181, :283-302) the convolution class and its initialisers: VarianceScaling(scale 2, fan_out, untruncated normal), zero biases"""
        if self.use_separable_conv:
            return SeparableConv2D(filters, kernel_size, padding="same", depthwise_initializer="he_normal_fan_out",
                                   pointwise_initializer="he_normal_fan_out", bias_initializer="zeros", trainable=self.trainable, name=name)
        return Conv2D(filters, kernel_size, padding="same", kernel_initializer="he_normal_fan_out", bias_initializer="zeros",
                      trainable=self.trainable, name=name)

    def _build_layers(self):
        nf = self.num_filters
        self.resample = torch.nn.ModuleDict()
        for level in range(self.min_level, self.max_level + 1):
            spec = self.input_specs.get(str(level))
            if spec is not None and int(spec[-1]) != nf:      # (:254-262) only where the channel count differs
                conv = self._conv(nf, 1, f"{self.name}/resample_l{level}/separable_conv2d")
                bn = BatchNormalization(name=f"{self.name}/resample_l{level}/bn", trainable=self.trainable, **self.norm_kwargs)
                conv.build((None, None, None, int(spec[-1])))
                bn.build((None, None, None, nf))
                conv.built = bn.built = True
                self.resample[str(level)] = torch.nn.ModuleList([conv, bn])
        self.cells = torch.nn.ModuleList()
        n_levels = self.max_level - self.min_level + 1
        for r in range(self.num_repeats):
            cell = torch.nn.ModuleList()
            for i in range(len(self.block_specs)):
                prefix = f"{self.name}/cell_{r}/sub_policy{i}/op_after_combine{n_levels + i}"
                conv = self._conv(nf, (3, 3), f"{prefix}/conv")
                bn = BatchNormalization(name=f"{prefix}/bn", trainable=self.trainable, **self.norm_kwargs)
                conv.build((None, None, None, nf))
                bn.build((None, None, None, nf))
                conv.built = bn.built = True
                cell.append(torch.nn.ModuleList([conv, bn]))
            self.cells.append(cell)
        self.built = True

    def build(self, input_shape):
        self.built = True

    @staticmethod
    def _resample(x, input_level, target_level):
        """(:264-271) inside a cell every node already has num_filters channels: only the resolution changes"""
        if input_level < target_level:
            stride = int(2 ** (target_level - input_level))
            return F.max_pool2d(x, stride, strides=stride, padding="same")
        if input_level > target_level:
            return nearest_upsampling(x, int(2 ** (input_level - target_level)))
        return x

    def _cell(self, feats, cell, training):
        feats = list(feats)
        levels = list(range(self.min_level, self.max_level + 1))
        used = [0] * len(feats)
        n_levels = len(levels)
        for i, spec in enumerate(self.block_specs):
            new_level = spec.level
            i0, i1 = spec.input_offsets
            if max(i0, i1) >= len(feats):
                raise ValueError(f"input_offset ({max(i0, i1)}) is larger than num feats({len(feats)})")
            node0, l0 = feats[i0], levels[i0]
            node1, l1 = feats[i1], levels[i1]
            used[i0] += 1
            used[i1] += 1
            node0 = self._resample(node0, l0, new_level)
            node1 = self._resample(node1, l1, new_level)
            if self.use_sum_for_combination or spec.combine_fn == "sum":
                new_node = F.add(node0, node1)
            elif spec.combine_fn == "attention":
                new_node = global_attention(node0, node1) if l0 >= l1 else global_attention(node1, node0)
            else:
                raise ValueError(f"unknown combine_fn `{spec.combine_fn}`.")
            if spec.is_output:      # (:348-355) every node of this level nobody has read yet joins the output
                for j in range(len(feats)):
                    if used[j] == 0 and levels[j] == new_level:
                        used[j] += 1
                        new_node = F.add(new_node, feats[j])
            conv, bn = cell[i]
            new_node = bn(conv(self.activation(new_node), training=training), training=training)
            feats.append(new_node)
            levels.append(new_level)
            used.append(0)
        return {levels[i]: feats[i] for i in range(len(feats) - n_levels, len(feats))}

    def call(self, inputs, training=None):
        inputs = {str(k): v for k, v in inputs.items()}
        feats = []
        for level in range(self.min_level, self.max_level + 1):
            key = str(level)
            if key in inputs:
                x = inputs[key]
                if key in self.resample:
                    conv, bn = self.resample[key]
                    x = bn(conv(x, training=training), training=training)
                feats.append(x)
            else:      # (:226-232) a level the backbone does not have: stride-2 max-pool of the level below
                feats.append(F.max_pool2d(feats[-1], 2, strides=2, padding="same"))
        out = None
        for r in range(self.num_repeats):
            out = self._cell(feats, self.cells[r], training)
            feats = [out[level] for level in range(self.min_level, self.max_level + 1)]
        return {str(level): out[level] for level in range(self.min_level, self.max_level + 1)}
