"""layers/model_builder.py of the reference: ConvNormAct (:34-115), ImageLevelBlock (:253-273), CommonEndBlock (:276-296),
get_training_value (:20-31) -- same constructor keywords, attribute names and call order, on HIP operators."""
import torch

from .. import functional as F
from .. import nn
from ..nn import Layer
from ..utils.common import resize_image
from .base_layers import Conv2D, Dropout, get_activation
from .normalizations import normalization


def get_training_value(training=None):
    if training is None:
        return False
    if isinstance(training, int):
        training = bool(training)
    return training


class ConvNormAct(Layer):
    def __init__(self, filters=256, kernel_size=1, dilation_rate=1, use_bn=True, activation="relu",
                 kernel_initializer="glorot_uniform", dropout_rate=0, dropout_before_bn=False, trainable=True, use_bias=False,
                 groups=1, conv_func=Conv2D, norm_func=normalization, name=None):
        super().__init__(trainable=trainable, name=name if name is not None else "ConvBnRelu")
        self.conv = conv_func(filters, kernel_size, padding="same", use_bias=use_bias, kernel_initializer=kernel_initializer,
                              dilation_rate=dilation_rate, groups=groups, trainable=trainable, name=f"{self.name}/conv")
        self.bn = None if not use_bn else norm_func(trainable=trainable, name=f"{self.name}/bn")
        self.activation = get_activation(activation)
        self.dropout = None
        self.dropout_before_bn = dropout_before_bn
        if dropout_rate > 0:
            self.dropout = Dropout(dropout_rate, name=f"{self.name}/dropout")

    def call(self, inputs, training=None):
        x = self.conv(inputs)
        should_dropout = (self.dropout is not None) and self.trainable
        if should_dropout and self.dropout_before_bn:
            x = self.dropout(x, training=training)
        act = self.activation
        if self.bn is not None:
            fuse = act is F.relu and hasattr(self.bn, "moving_mean")   # BN + ReLU in one pass over the activation
            x = self.bn(x, training=training, fused_relu=True) if fuse else self.bn(x, training=training)
            if fuse:
                act = None
        if act is not None:
            x = act(x)
        if should_dropout and not self.dropout_before_bn:
            x = self.dropout(x, training=training)
        return x

    def reset_weights(self):
        """layers/model_builder.py:100-115: every variable of the block goes back to a fresh draw of its initializer"""
        _reset_weight(self.conv.kernel, self.conv.kernel_initializer)
        if self.conv.use_bias:
            _reset_weight(self.conv.bias, self.conv.bias_initializer)
        if self.bn is not None:
            _reset_weight(self.bn.beta, self.bn.beta_initializer)
            _reset_weight(self.bn.gamma, self.bn.gamma_initializer)
            _reset_weight(self.bn.moving_mean, self.bn.moving_mean_initializer)
            _reset_weight(self.bn.moving_variance, self.bn.moving_variance_initializer)


_RESET_DRAWS = [0]


def _reset_weight(weight, initializer):
    """weight.assign(initializer(shape, dtype)) on a view of the flat parameter buffer; the bf16 compute copy follows"""
    from ..nn import init_tensor

    if weight is None:
        return
    _RESET_DRAWS[0] += 1      # a new draw each time, like a Keras initializer object without a fixed seed
    name = f"{getattr(weight, 'iseg_name', 'w')}#reset{_RESET_DRAWS[0]}"
    fresh = init_tensor(initializer, tuple(weight.shape), name).to(weight.device)
    with torch.no_grad():
        (weight.data if isinstance(weight, torch.nn.Parameter) else weight).copy_(fresh)
        shadow = getattr(weight, "iseg_compute", None)
        if shadow is not None:
            shadow.copy_(fresh.to(shadow.dtype))
            nn.weights_changed()


class ImageLevelBlock(Layer):
    def __init__(self, filters=256, pooling_axis=(1, 2), name=None):
        super().__init__(name="ImageLevelBlock" if name is None else name)
        self.convbnrelu = ConvNormAct(filters, (1, 1), name=f"{self.name}/conv")
        self.pooling_axis = tuple(pooling_axis)
        if self.pooling_axis != (1, 2):
            raise NotImplementedError("ImageLevelBlock: pooling over (H, W) only")

    def call(self, inputs, training=None):
        h, w = inputs.shape[1], inputs.shape[2]
        x = F.global_avg_pool(inputs)
        x = self.convbnrelu(x, training=training)
        return F.broadcast_hw(x, h, w)


class CommonEndBlock(Layer):
    def __init__(self, filters=256, num_class=21, dropout_rate=0.1, name=None):
        super().__init__(name=name)
        self.filters, self.num_class, self.dropout_rate = filters, num_class, dropout_rate
        self.end_conv = ConvNormAct(self.filters, dropout_rate=self.dropout_rate, name=f"{self.name}/end_conv")
        self.logits_conv = Conv2D(self.num_class, (1, 1), name=f"{self.name}/logits_conv")

    def call(self, inputs, training=False):
        x, orginal_inputs = inputs
        x = self.end_conv(x, training=training)
        x = self.logits_conv(x)
        x = resize_image(x, orginal_inputs.shape[1:3])
        return F.cast_to(x, torch.float32)


class SepConvBnReLU(Layer):
    """layers/model_builder.py:118-171: depthwise conv (no bias) -> [BN] -> activation -> [pointwise ConvNormAct]"""

    def __init__(self, filters, kernel_size, apply_bn=True, dilation_rate=1, activation="relu", apply_pointwise=True,
                 apply_pointwise_bn=True, name=None):
        super().__init__(name=name)
        from .base_layers import DepthwiseConv2D

        self.use_bn = apply_bn
        self.activation = get_activation(activation)
        self.apply_pointwise = apply_pointwise
        self.depthwise_conv = DepthwiseConv2D(kernel_size, padding="same", dilation_rate=dilation_rate, use_bias=False,
                                              name=f"{self.name}/depthwise_conv")
        if self.use_bn:
            self.depthwise_bn = normalization(name=f"{self.name}/depthwise_bn")
        if self.apply_pointwise:
            self.pointwise_conv = ConvNormAct(filters, use_bn=apply_pointwise_bn, activation=activation, name=f"{self.name}/pointwise_conv")

    def call(self, inputs, training=None):
        x = self.depthwise_conv(inputs)
        act = self.activation
        if self.use_bn:
            fuse = act is F.relu and hasattr(self.depthwise_bn, "moving_mean")
            x = self.depthwise_bn(x, training=training, fused_relu=True) if fuse else self.depthwise_bn(x, training=training)
            if fuse:
                act = None
        if act is not None:
            x = act(x)
        if self.apply_pointwise:
            x = self.pointwise_conv(x, training=training)
        return x


class NormConvAct(Layer):
    """layers/model_builder.py:175-250: [LN | GN | BN | RMSN] -> Conv2D(padding same, activation)"""

    def __init__(self, filters=256, kernel_size=1, dilation_rate=(1, 1), use_norm=True, norm_type="ln", ln_epsilon=1e-6, activation="gelu",
                 conv_kernel_initializer="glorot_uniform", dropout_rate=0, trainable=True, use_bias=True, groups=1, name=None):
        super().__init__(trainable=trainable, name=name)
        from .. import static_strings as ss
        from .base_layers import BatchNormalization, LayerNormalization
        from .groupnorm import GroupNormalization
        from .rmsnorm import RMSNormalization

        self.ln = None
        if use_norm:
            if norm_type == ss.BN:
                self.ln = BatchNormalization(trainable=trainable, epsilon=ln_epsilon, synchronized=True, name=f"{self.name}_bn")
            elif norm_type in (ss.LN, ss.GN):
                if groups == 1:
                    self.ln = LayerNormalization(trainable=trainable, epsilon=ln_epsilon, name=f"{self.name}_ln")
                elif groups > 1:
                    self.ln = GroupNormalization(groups=groups, axis=-1, epsilon=ln_epsilon, trainable=trainable, name=f"{self.name}_ln")
                else:
                    raise ValueError(f"Invalid groups value: {groups}")
            elif norm_type == ss.RMSN:
                self.ln = RMSNormalization(epsilon=ln_epsilon, trainable=trainable, name=f"{self.name}_rmsn")
            else:
                raise ValueError(f"Invalid norm_type: {norm_type}")
        self.conv = Conv2D(filters, kernel_size, padding="same", use_bias=use_bias, kernel_initializer=conv_kernel_initializer,
                           dilation_rate=dilation_rate, trainable=trainable, activation=activation, name=f"{self.name}_conv")

    def call(self, inputs, training=None):
        x = inputs
        if self.ln is not None:      # (Keras hands the enclosing call's `training` to a nested BatchNormalization)
            x = self.ln(x, training=training) if hasattr(self.ln, "moving_mean") else self.ln(x)
        return self.conv(x)

    def reset_weights(self):
        """layers/model_builder.py:234-247"""
        _reset_weight(self.conv.kernel, self.conv.kernel_initializer)
        if self.conv.use_bias:
            _reset_weight(self.conv.bias, self.conv.bias_initializer)
        if self.ln is not None:
            _reset_weight(getattr(self.ln, "beta", None), getattr(self.ln, "beta_initializer", "zeros"))
            _reset_weight(getattr(self.ln, "gamma", None), getattr(self.ln, "gamma_initializer", "ones"))
