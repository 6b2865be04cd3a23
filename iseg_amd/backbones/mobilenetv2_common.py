"""MobileNetV2 (backbones/mobilenetv2_common.py of the reference): MobileNetV2 :16-81, InvertedResBlock :84-176, correct_pad :179-190,
_make_divisible :193-201, build_atrous_mobilenetv2 :204-222 -- same classes, attributes (`strides` / `atrous_rates` properties the dilation
surgery edits) and weight names (Conv1, bn_Conv1, expanded_conv_*, block_<id>_{expand,depthwise,project}[_BN], Conv_1, Conv_1_bn).

A stride-2 block of the reference zero-pads by correct_pad and runs its 3x3 depthwise convolution with padding "valid"; that is exactly the
'same' stride-2 sampling for a 3x3 kernel (even sizes pad (0, 1), odd ones (1, 1)), so the block calls the strided 'same' depthwise operator
(functional.depthwise_conv2d(strides=2)) and no padded copy of the activation exists."""
from .. import functional as F
from ..layers.base_layers import Conv2D, DepthwiseConv2D
from ..layers.normalizations import normalization
from ..nn import Layer


def _make_divisible(v, divisor, min_value=None):
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:      # rounding down must not lose more than 10 %
        new_v += divisor
    return new_v


def correct_pad(inputs_shape, kernel_size):
    """((top, bottom), (left, right)) of the reference's ZeroPadding2D in front of a stride-2 'valid' convolution (kept for API parity)"""
    size = inputs_shape[1:3]
    if isinstance(kernel_size, int):
        kernel_size = (kernel_size, kernel_size)
    adjust = (1, 1) if size[0] is None else (1 - size[0] % 2, 1 - size[1] % 2)
    correct = (kernel_size[0] // 2, kernel_size[1] // 2)
    return ((correct[0] - adjust[0], correct[0]), (correct[1] - adjust[1], correct[1]))


def _bn_relu6(bn, x, training):
    return F.relu6(bn(x, training=training))


class InvertedResBlock(Layer):
    def __init__(self, expansion, stride, alpha, filters, block_id):
        super().__init__(name=f"block_{block_id}" if block_id else "expanded_conv")
        self.expansion = expansion
        self.orginal_stride = stride
        self.alpha = alpha
        self.filters = filters
        self.block_id = block_id
        self.prefix = "block_{}_".format(block_id) if block_id else "expanded_conv_"
        self.depthwise = DepthwiseConv2D((3, 3), strides=stride, use_bias=False, padding="same", name=self.prefix + "depthwise")
        self.depthwise_bn = normalization(name=self.prefix + "depthwise_BN")
        self.expand_conv = self.expand_bn = None

    def build(self, input_shape):
        self.in_channels = int(input_shape[-1])
        self.pointwise_filters = _make_divisible(int(self.filters * self.alpha), 8)
        if self.block_id:
            self.expand_conv = Conv2D(self.expansion * self.in_channels, (1, 1), padding="same", use_bias=False, name=self.prefix + "expand")
            self.expand_bn = normalization(name=self.prefix + "expand_BN")
        self.project = Conv2D(self.pointwise_filters, (1, 1), padding="same", use_bias=False, name=self.prefix + "project")
        self.project_bn = normalization(name=self.prefix + "project_BN")
        self.built = True

    @property
    def strides(self):
        return self.depthwise.strides[0]

    @strides.setter
    def strides(self, value):
        self.depthwise.strides = (int(value), int(value)) if not isinstance(value, (tuple, list)) else tuple(value)

    @property
    def atrous_rates(self):
        return self.depthwise.dilation_rate[0]

    @atrous_rates.setter
    def atrous_rates(self, value):
        self.depthwise.dilation_rate = (int(value), int(value)) if not isinstance(value, (tuple, list)) else tuple(value)

    def call(self, inputs, training=None):
        residual = self.in_channels == self.pointwise_filters and self.orginal_stride == 1
        if residual:
            x, skip = F.fork(inputs, 2)
        else:
            x, skip = inputs, None
        if self.block_id:
            x = _bn_relu6(self.expand_bn, self.expand_conv(x), training)
        x = _bn_relu6(self.depthwise_bn, self.depthwise(x), training)
        x = self.project_bn(self.project(x), training=training)
        return F.add(skip, x) if residual else x


class MobileNetV2(Layer):
    def __init__(self, alpha=1.0, return_endpoints=False, name=None):
        super().__init__(name=name)
        import torch

        self.conv1 = Conv2D(_make_divisible(32 * alpha, 8), (3, 3), strides=(2, 2), padding="same", use_bias=False, name="Conv1")
        self.bn_conv1 = normalization(name="bn_Conv1")
        self.blocks = torch.nn.ModuleList()
        self.__add_blocks(16, alpha, stride=1, expansion=1, repeated=1)
        self.__add_blocks(24, alpha, stride=2, expansion=6, repeated=2)
        self.__add_blocks(32, alpha, stride=2, expansion=6, repeated=3)
        self.__add_blocks(64, alpha, stride=2, expansion=6, repeated=4)
        self.__add_blocks(96, alpha, stride=1, expansion=6, repeated=3)
        self.__add_blocks(160, alpha, stride=2, expansion=6, repeated=3)
        self.__add_blocks(320, alpha, stride=1, expansion=6, repeated=1)
        last = _make_divisible(1280 * alpha, 8) if alpha > 1.0 else 1280
        self.last_block_conv = Conv2D(last, (1, 1), use_bias=False, name="Conv_1")
        self.last_block_conv_bn = normalization(name="Conv_1_bn")
        self.return_endpoints = return_endpoints

    def __add_blocks(self, filters, alpha, stride=1, expansion=1, repeated=1):
        for i in range(repeated):
            self.blocks.append(InvertedResBlock(filters=filters, alpha=alpha, stride=stride if i == 0 else 1, expansion=expansion,
                                                block_id=len(self.blocks)))

    def call(self, inputs, training=None):
        endpoints = []
        x = _bn_relu6(self.bn_conv1, self.conv1(F.cast_input(inputs)), training)
        for block in self.blocks:
            if block.orginal_stride > 1:
                if self.return_endpoints:
                    x, e = F.fork(x, 2)
                    endpoints.append(e)
            x = block(x, training=training)
        x = _bn_relu6(self.last_block_conv_bn, self.last_block_conv(x), training)
        endpoints.append(x)
        return endpoints if self.return_endpoints else x


def build_atrous_mobilenetv2(net, output_stride=32):
    current_os = 2
    current_dilation_rate = 1
    for block in net.blocks:
        if block.strides > 1:
            if current_os >= output_stride:
                current_dilation_rate *= block.strides
                block.strides = 1
                block.atrous_rates = current_dilation_rate
            else:
                current_os *= block.strides
        else:
            block.atrous_rates = current_dilation_rate
    return net
