"""ConvNeXt (backbones/convnext.py of the reference): Block :16-63, DownSampleLayer :66-91, Stage :94-125,
ConvNeXt :128-201, convnext_* :205-242, build_dilated_convnext :245-266 -- same classes, attributes and weight names;
a Block's forward+backward is ONE fused operator (functional.convnext_block) over the sub-layers' variables."""
import types

import numpy as np
import torch

from .. import functional as F
from ..layers.base_layers import Conv2D, Dense, DepthwiseConv2D, LayerNormalization
from ..nn import Layer


class Block(Layer):
    def __init__(self, filters, drop_path_prob=0.0, layer_scale_init_value=1e-6, name=None):
        super().__init__(name=name)
        self.drop_path_prob = float(drop_path_prob)
        self.layer_scale_init_value = layer_scale_init_value
        self.filters = filters
        self.dwconv = DepthwiseConv2D(kernel_size=7, padding="same", name=f"{self.name}/dwconv")
        self.norm = LayerNormalization(epsilon=1e-6, name=f"{self.name}/norm")
        self.pwconv1 = Dense(units=4 * filters, name=f"{self.name}/pwconv1")
        self.pwconv2 = Dense(units=filters, name=f"{self.name}/pwconv2")
        self.gamma = None
        self.drop_path_mask = None   # parity tests may inject the per-sample factors

    def build(self, input_shape):
        c = self.filters
        self.dwconv.build((None, None, None, c))
        self.norm.build((None, None, None, c))
        self.pwconv1.build((None, None, None, c))
        self.pwconv2.build((None, None, None, 4 * c))
        if self.layer_scale_init_value > 0:
            self.gamma = self.add_weight("gamma", (c,), float(self.layer_scale_init_value))
        self.built = True

    def _params(self):
        return types.SimpleNamespace(dw_kernel=self.dwconv.depthwise_kernel, dw_bias=self.dwconv.bias, ln_gamma=self.norm.gamma,
                                     ln_beta=self.norm.beta, w1=self.pwconv1.kernel, b1=self.pwconv1.bias, w2=self.pwconv2.kernel,
                                     b2=self.pwconv2.bias, gamma=self.gamma)

    def call(self, inputs, training=None):
        mask = None
        if self.drop_path_prob != 0.0 and training:
            mask = self.drop_path_mask
            if mask is None:
                mask = F.drop_path_factors(inputs.shape[0], 1.0 - self.drop_path_prob, inputs.device)
        d = self.dwconv.dilation_rate
        return F.convnext_block(inputs, self._params(), d[0], self.norm.epsilon, mask)


class DownSampleLayer(Layer):
    def __init__(self, filters=96, strides=2, swap=False, name=None):
        super().__init__(name=name)
        self.swap = swap
        names = ["1", "0"] if swap else ["0", "1"]
        self.norm = LayerNormalization(epsilon=1e-6, name=f"{self.name}/{names[0]}")
        self.conv = Conv2D(filters=filters, kernel_size=strides, strides=strides, padding="same", name=f"{self.name}/{names[1]}")

    def call(self, inputs, training=None):
        x = inputs
        if self.swap:
            x = self.norm(self.conv(x))
        else:
            x = self.conv(self.norm(x))
        return x


class Stage(Layer):
    def __init__(self, filters=96, depth=3, drop_path_probs=[], layer_scale_init_value=1e-6, name=None):
        super().__init__(name=name)
        assert len(drop_path_probs) == 0 or len(drop_path_probs) == depth
        self.blocks = torch.nn.ModuleList([
            Block(filters=filters, drop_path_prob=drop_path_probs[i], layer_scale_init_value=layer_scale_init_value,
                  name=f"{self.name}/{i}") for i in range(depth)
        ])

    def call(self, inputs, training=None):
        x = inputs
        for block in self.blocks:
            x = block(x, training=training)
        return x


class ConvNeXt(Layer):
    def __init__(self, depths=[3, 3, 9, 3], filters_list=[96, 192, 384, 768], drop_path_rate=0.0, layer_scale_init_value=1e-6,
                 return_endpoints=False, name=None):
        super().__init__(name=name)
        self.return_endpoints = return_endpoints
        num_stage = len(depths)
        assert num_stage == len(filters_list)
        drop_path_rates = np.linspace(0.0, drop_path_rate, sum(depths))
        downs, stages = [], []
        cur = 0
        for i in range(num_stage):
            downs.append(DownSampleLayer(filters=filters_list[i], strides=4 if i == 0 else 2, swap=i == 0,
                                         name=f"downsample_layers/{i}"))
            stages.append(Stage(filters=filters_list[i], depth=depths[i], drop_path_probs=drop_path_rates[cur:cur + depths[i]],
                                layer_scale_init_value=layer_scale_init_value, name=f"stages/{i}"))
            cur += depths[i]
        self.downsample_blocks = torch.nn.ModuleList(downs)
        self.stages = torch.nn.ModuleList(stages)

    def call(self, inputs, training=None):
        # the 4x4 / stride-4 stem has 3 input channels, i.e. it always takes the im2col route, and the patch kernel rounds fp32 -> bf16 itself:
        # a separate cast pass over the image (25 us at 16 x 512 x 512) would write and re-read it for nothing
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


def convnext_tiny(return_endpoints=False):
    return ConvNeXt(depths=[3, 3, 9, 3], filters_list=[96, 192, 384, 768], return_endpoints=return_endpoints, drop_path_rate=0.1)


def convnext_large(return_endpoints=False):
    return ConvNeXt(depths=[3, 3, 27, 3], filters_list=[192, 384, 768, 1536], return_endpoints=return_endpoints, drop_path_rate=0.3)


def convnext_xlarge(return_endpoints=False):
    return ConvNeXt(depths=[3, 3, 27, 3], filters_list=[256, 512, 1024, 2048], return_endpoints=return_endpoints, drop_path_rate=0.4)


def convnext_xxlarge(return_endpoints=False):
    return ConvNeXt(depths=[3, 4, 30, 3], filters_list=[384, 768, 1536, 3072], return_endpoints=return_endpoints, drop_path_rate=0.4)


def build_dilated_convnext(model: ConvNeXt, output_stride=32):
    num_stages = len(model.stages)
    current_os = 1
    current_dilation = 1
    for i in range(num_stages):
        if current_os >= output_stride:
            current_dilation *= model.downsample_blocks[i].conv.strides[0]
            model.downsample_blocks[i].conv.strides = (1, 1)
            model.downsample_blocks[i].conv.dilation_rate = (current_dilation, current_dilation)
            for block in model.stages[i].blocks:
                block.dwconv.strides = (1, 1)
                block.dwconv.dilation_rate = (current_dilation, current_dilation)
        else:
            current_os *= model.downsample_blocks[i].conv.strides[0]
    return model
