"""ConvNeXt V2 (backbones/convnext_v2.py of the reference): GlobalResponseNormlizationLayer :17-60, Block :63-98, DownSampleLayer :101-126,
Stage :129-158, ConvNeXtV2 :161-233, convnext_v2_* :236-281 -- same classes, attributes, weight names and shapes (GRN gamma / beta are
[1, 1, 1, 4*filters], zero-initialised).  A V2 block has no layer scale; the response normalisation sits between the GELU and the second
pointwise product and is one operator here (functional.grn -> csrc/grn.hip), its per-(sample, channel) statistics kept in fp32.
build_dilated_convnext (backbones/convnext.py:245-266) applies unchanged: the attributes it edits are the same."""
import os
import types

import numpy as np
import torch

from .. import functional as F
from ..layers.base_layers import Dense, DepthwiseConv2D, LayerNormalization
from ..nn import Layer
from .convnext import DownSampleLayer, build_dilated_convnext  # noqa: F401  (the reference's V2 module carries the same surgery, :284-306)


class GlobalResponseNormlizationLayer(Layer):
    """(the class keeps the reference's spelling)"""

    def __init__(self, trainable=True, epsilon=1e-6, name=None):
        super().__init__(name=name, trainable=trainable)
        self.epsilon = epsilon
        self.gamma = self.beta = None

    def build(self, input_shape):
        channels = int(input_shape[-1])
        self.gamma = self.add_weight("gamma", (1, 1, 1, channels), "zeros")
        self.beta = self.add_weight("beta", (1, 1, 1, channels), "zeros")
        self.built = True

    def call(self, inputs, training=None):
        return F.grn(inputs, self.gamma, self.beta, self.epsilon)


class Block(Layer):
    def __init__(self, filters, drop_path_prob=0.0, name=None):
        super().__init__(name=name)
        self.drop_path_prob = float(drop_path_prob)
        self.filters = filters
        self.dwconv = DepthwiseConv2D(kernel_size=7, padding="same", name=f"{self.name}/dwconv")
        self.norm = LayerNormalization(epsilon=1e-6, name=f"{self.name}/norm")
        self.pwconv1 = Dense(units=4 * filters, activation="gelu", name=f"{self.name}/pwconv1")      # tf.nn.gelu rides the product's epilogue
        self.grn = GlobalResponseNormlizationLayer(trainable=self.trainable, name=f"{self.name}/grn")
        self.pwconv2 = Dense(units=filters, name=f"{self.name}/pwconv2")
        self.drop_path_mask = None   # parity tests may inject the per-sample factors

    def build(self, input_shape):
        c = self.filters
        self.dwconv.build((None, None, None, c))
        self.norm.build((None, None, None, c))
        self.pwconv1.build((None, None, None, c))
        self.grn.build((None, None, None, 4 * c))
        self.pwconv2.build((None, None, None, 4 * c))
        self.built = True

    def _params(self):
        return types.SimpleNamespace(dw_kernel=self.dwconv.depthwise_kernel, dw_bias=self.dwconv.bias, ln_gamma=self.norm.gamma,
                                     ln_beta=self.norm.beta, w1=self.pwconv1.kernel, b1=self.pwconv1.bias, grn_gamma=self.grn.gamma,
                                     grn_beta=self.grn.beta, w2=self.pwconv2.kernel, b2=self.pwconv2.bias)

    def call(self, inputs, training=None):
        mask = None
        if self.drop_path_prob != 0.0 and training:
            mask = self.drop_path_mask
            if mask is None:
                mask = F.drop_path_factors(inputs.shape[0], 1.0 - self.drop_path_prob, inputs.device)
        if os.environ.get("ISEG_V2_BLOCK_FUSED", "1") == "0":
            return self.call_layers(inputs, mask)
        d = self.dwconv.dilation_rate
        return F.convnext_v2_block(inputs, self._params(), d[0], self.norm.epsilon, self.grn.epsilon, mask)      # one tape node per block

    def call_layers(self, inputs, mask):
        """the same block layer by layer through the generic operators (ISEG_V2_BLOCK_FUSED=0; the tests hold the two against each other)"""
        x, skip = F.fork(inputs, 2)      # residual fork: the two gradients are summed by our own kernel
        x = self.dwconv(x)
        x = self.norm(x)
        x = self.pwconv1(x)
        x = self.grn(x)
        x = self.pwconv2(x)
        if mask is not None:
            x = F.drop_path(x, self.drop_path_prob, True, mask=mask)
        return F.add(x, skip)


class Stage(Layer):
    def __init__(self, filters=96, depth=3, drop_path_probs=[], name=None):
        super().__init__(name=name)
        assert len(drop_path_probs) == 0 or len(drop_path_probs) == depth
        self.blocks = torch.nn.ModuleList([
            Block(filters=filters, drop_path_prob=drop_path_probs[i], name=f"{self.name}/{i}") for i in range(depth)
        ])

    def call(self, inputs, training=None):
        x = inputs
        for block in self.blocks:
            x = block(x, training=training)
        return x


class ConvNeXtV2(Layer):
    def __init__(self, depths=[3, 3, 9, 3], filters_list=[96, 192, 384, 768], drop_path_rate=0.0, return_endpoints=False, name=None):
        super().__init__(name=name)
        self.return_endpoints = return_endpoints
        num_stage = len(depths)
        assert num_stage == len(filters_list)
        drop_path_rates = np.linspace(0.0, drop_path_rate, sum(depths))
        downs, stages = [], []
        cur = 0
        for i in range(num_stage):
            downs.append(DownSampleLayer(filters=filters_list[i], strides=4 if i == 0 else 2, swap=i == 0, name=f"downsample_layers/{i}"))
            stages.append(Stage(filters=filters_list[i], depth=depths[i], drop_path_probs=drop_path_rates[cur:cur + depths[i]],
                                name=f"stages/{i}"))
            cur += depths[i]
        self.downsample_blocks = torch.nn.ModuleList(downs)
        self.stages = torch.nn.ModuleList(stages)

    def call(self, inputs, training=None):
        # (the 3-channel stem takes the patch route, which rounds fp32 -> bf16 itself: see ConvNeXt.call)
        x = inputs if (torch.is_tensor(inputs) and inputs.dtype == torch.float32 and inputs.shape[-1] % 8 != 0) else F.cast_input(inputs)
        endpoints = [None]
        for i in range(len(self.stages)):
            x = self.downsample_blocks[i](x, training=training)
            x = self.stages[i](x, training=training)
            endpoints += [x]
        if self.return_endpoints:
            return endpoints
        return x

    def decay_lr(self, rate=0.99):
        from .utils.layerwise_decay import decay_layers_lr

        stages = list(self.stages)
        stages.reverse()
        decay_layers_lr(stages, rate=rate)


def convnext_v2_nano(return_endpoints=False):
    return ConvNeXtV2(depths=[2, 2, 8, 2], filters_list=[80, 160, 320, 640], return_endpoints=return_endpoints, drop_path_rate=0.1)


def convnext_v2_tiny(return_endpoints=False):
    return ConvNeXtV2(depths=[3, 3, 9, 3], filters_list=[96, 192, 384, 768], return_endpoints=return_endpoints, drop_path_rate=0.1)


def convnext_v2_large(return_endpoints=False):
    return ConvNeXtV2(depths=[3, 3, 27, 3], filters_list=[192, 384, 768, 1536], return_endpoints=return_endpoints, drop_path_rate=0.3)


def convnext_v2_huge(return_endpoints=False):
    return ConvNeXtV2(depths=[3, 3, 27, 3], filters_list=[352, 704, 1408, 2816], return_endpoints=return_endpoints, drop_path_rate=0.4)
