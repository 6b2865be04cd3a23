"""TensorFlow / Keras adapter for tests/golden/make_golden.py --impl tf: the op vocabulary of oracle/tf_ops.py and oracle/models.py
computed by the REAL thing, so that the committed fixture script can regenerate every golden vector from TensorFlow on the same seeded
inputs (and, for the ops that exist only in the reference -- DCNv3, the Swin tables, WarmUpPolyDecay, AdamW_EXT, the sliding-window
indices, RMSNormalization -- from a checkout of the reference given with --reference).

THIS FILE CANNOT RUN IN THE BUILD IMAGE (no TensorFlow, no network); it is written against the public TF >= 2.10 / Keras 2 APIs the
reference itself uses and is meant to be run once, off-box, by whoever has that environment.  Functions take / return torch float64
tensors like the oracle's.  An op whose backend is missing returns None and is left out of the fixture."""
import math
import sys

import numpy as np
import torch


def _tf():
    import tensorflow as tf

    tf.keras.backend.set_floatx("float64")
    return tf


def _to(x):
    return None if x is None else (x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x))


def _back(y):
    return torch.from_numpy(np.asarray(y.numpy() if hasattr(y, "numpy") else y))


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class _Ops:
    def __init__(self, reference=None):
        self.ref = reference
        if reference:
            import os

            parent = os.path.dirname(os.path.abspath(reference))
            if parent not in sys.path:
                sys.path.insert(0, parent)      # the reference imports itself as `iseg.*`: the checkout must be a directory named iseg

    # ---- Keras / TF level ------------------------------------------------------------------------------------------------------
    def conv2d(self, x, kernel, bias=None, strides=1, dilation=1, padding="same", groups=1):
        tf = _tf()
        k = _to(kernel)
        layer = tf.keras.layers.Conv2D(k.shape[-1], k.shape[:2], strides=_pair(strides), padding=padding, dilation_rate=_pair(dilation),
                                       groups=groups, use_bias=bias is not None)
        xx = tf.constant(_to(x))
        layer.build(xx.shape)
        layer.set_weights([k] + ([_to(bias)] if bias is not None else []))
        return _back(layer(xx))

    def depthwise_conv2d(self, x, kernel, bias=None, strides=1, dilation=1, padding="same"):
        tf = _tf()
        k = _to(kernel)
        layer = tf.keras.layers.DepthwiseConv2D(k.shape[:2], strides=_pair(strides), padding=padding, dilation_rate=_pair(dilation),
                                                use_bias=bias is not None)
        xx = tf.constant(_to(x))
        layer.build(xx.shape)
        layer.set_weights([k] + ([_to(bias)] if bias is not None else []))
        return _back(layer(xx))

    def layer_norm(self, x, gamma, beta, eps):
        tf = _tf()
        layer = tf.keras.layers.LayerNormalization(axis=-1, epsilon=eps)
        xx = tf.constant(_to(x))
        layer.build(xx.shape)
        layer.set_weights([_to(gamma), _to(beta)])
        return _back(layer(xx))

    def gelu(self, x):
        return _back(_tf().keras.activations.gelu(_tf().constant(_to(x))))

    def resize_bilinear(self, x, size):
        tf = _tf()
        return _back(tf.cast(tf.image.resize(tf.constant(_to(x)), list(size), method="bilinear"), tf.float64))

    def resize_nearest(self, x, size):
        tf = _tf()
        return _back(tf.image.resize(tf.constant(_to(x)), list(size), method="nearest"))

    def resize_bicubic(self, x, size):
        tf = _tf()
        return _back(tf.cast(tf.image.resize(tf.constant(_to(x)), list(size), method="bicubic"), tf.float64))

    def softmax_ce_ignore(self, y_true, logits, num_class=21, ignore_label=255, class_weights=None):
        if self.ref:      # the reference's own loss factory
            from iseg.losses.catecrossentropy_ignore_label import catecrossentropy_ignore_label_loss

            fn = catecrossentropy_ignore_label_loss(num_class=num_class, ignore_label=ignore_label, class_weights=class_weights)
            return _back(fn(_tf().constant(_to(y_true)), _tf().constant(_to(logits))))
        tf = _tf()
        y = tf.reshape(tf.constant(_to(y_true)), [-1])
        z = tf.reshape(tf.constant(_to(logits)), [-1, num_class])
        w = tf.cast(tf.not_equal(y, ignore_label), z.dtype)
        if ignore_label == 0:
            y = y - 1
        onehot = tf.one_hot(tf.cast(y, tf.int32), num_class, dtype=z.dtype)
        if class_weights is not None and len(class_weights) > 0:
            w = w * tf.reduce_sum(onehot * tf.constant(np.asarray(class_weights), z.dtype), -1)
        loss = tf.keras.losses.CategoricalCrossentropy(from_logits=True, reduction=tf.keras.losses.Reduction.NONE)
        return _back(loss(onehot, z, sample_weight=w))

    def argmax_first(self, logits):
        return _back(_tf().argmax(_tf().constant(_to(logits)), axis=-1))

    def confusion_matrix(self, labels, preds, num_class, ignore_label):
        tf = _tf()
        y, p = tf.reshape(tf.constant(_to(labels)), [-1]), tf.reshape(tf.constant(_to(preds)), [-1])
        w = tf.cast(tf.not_equal(y, ignore_label), tf.float64)
        y = tf.where(tf.equal(y, ignore_label), tf.zeros_like(y), y)
        return _back(tf.math.confusion_matrix(y, p, num_classes=num_class, weights=w, dtype=tf.float64))

    def batch_norm_train(self, x, gamma, beta, eps, stats=None):
        tf = _tf()
        layer = tf.keras.layers.BatchNormalization(epsilon=eps, momentum=0.9)
        xx = tf.constant(_to(x))
        layer.build(xx.shape)
        layer.set_weights([_to(gamma), _to(beta), np.zeros(xx.shape[-1]), np.ones(xx.shape[-1])])
        y = layer(xx, training=True)
        mean, var = tf.nn.moments(xx, axes=[0, 1, 2])
        return _back(y), _back(mean), _back(var)

    def group_norm(self, x, gamma, beta, groups, eps=1e-3):
        tf = _tf()
        layer = tf.keras.layers.GroupNormalization(groups=groups, axis=-1, epsilon=eps, center=beta is not None, scale=gamma is not None)
        xx = tf.constant(_to(x))
        layer.build(xx.shape)
        if gamma is not None:
            layer.set_weights([_to(gamma), _to(beta)])
        return _back(layer(xx))

    def rms_norm(self, x, scale, eps=1e-6):
        if not self.ref:
            return None
        from iseg.layers.rmsnorm import RMSNormalization

        layer = RMSNormalization(epsilon=eps)
        xx = _tf().constant(_to(x))
        layer.build(xx.shape)
        layer.set_weights([_to(scale)])
        return _back(layer(xx))

    def max_pool_same(self, x, k, s):
        return _back(_tf().keras.layers.MaxPool2D(k, s, padding="same")(_tf().constant(_to(x))))

    def avg_pool_same(self, x, k, s):
        return _back(_tf().keras.layers.AveragePooling2D(k, s, padding="same")(_tf().constant(_to(x))))

    # ---- reference level -------------------------------------------------------------------------------------------------------
    def dcnv3_op(self, x, offset, mask, kernel_size=(3, 3), strides=(1, 1), padding="SAME", dilation_rate=(1, 1), groups=4, group_channels=16,
                 offset_scale=1.0):
        if not self.ref:
            return None
        from iseg.layers.dcn_v3.op import dcnv3_op

        tf = _tf()
        return _back(dcnv3_op(tf.constant(_to(x)), tf.constant(_to(offset)), tf.constant(_to(mask)), kernel_size=kernel_size, strides=strides,
                              padding=padding, dilation_rate=dilation_rate, groups=groups, group_channels=group_channels,
                              offset_scale=offset_scale))

    def warmup_poly_decay(self, step, initial_lr, decay_steps, end_lr=0.0001, warmup_steps=0, warmup_lr=1e-4, power=1.0):
        if not self.ref:
            return None
        from iseg.optimizers.polydecay import WarmUpPolyDecay

        return float(WarmUpPolyDecay(initial_lr, decay_steps, end_learning_rate=end_lr, power=power, warmup_steps=warmup_steps,
                                     warmup_learning_rate=warmup_lr)(step))

    def sliding_start_indexs(self, length, crop):
        if not self.ref:
            return None
        from iseg.utils.sliding_window_inference_utils import get_sliding_start_indexs

        return [int(v) for v in np.asarray(get_sliding_start_indexs(length, crop))]

    def adamw_step(self, w, g, m, v, step, lr, lr_mult=1.0, wd=0.0, beta1=0.9, beta2=0.999, eps=1e-7, vhat=None):
        """one step of the reference's AdamW_EXT on a single variable; optimizer state is rebuilt from (m, v, step) each call"""
        if not self.ref:
            return None
        from iseg.optimizers.modern.adamw import AdamW_EXT

        tf = _tf()
        var = tf.Variable(_to(w))
        opt = AdamW_EXT(learning_rate=lr, weight_decay=wd, beta_1=beta1, beta_2=beta2, epsilon=eps, amsgrad=vhat is not None)
        opt.build([var])
        opt.iterations.assign(step - 1)
        opt._momentums[0].assign(_to(m))
        opt._velocities[0].assign(_to(v))
        opt.apply_gradients([(tf.constant(_to(g)), var)])
        return _back(var), _back(opt._momentums[0]), _back(opt._velocities[0])


class _Models:
    def __init__(self, reference=None):
        self.ref = reference

    def swin_attention_mask(self, h, w, window, shift):
        if not self.ref:
            return None
        from iseg.backbones.swin import BasicLayer

        layer = BasicLayer.__new__(BasicLayer)
        layer.window_size, layer.shift_size = window, shift
        pad_h = int(math.ceil(h / window)) * window
        pad_w = int(math.ceil(w / window)) * window
        return _back(layer.generate_attention_mask(pad_h, pad_w))

    def _rel_index(self, window):
        if not self.ref:
            return None
        from iseg.backbones.swin import WindowAttention

        wa = WindowAttention(dim=32, window_size=(window, window), num_heads=1)
        wa.build((None, window * window, 32))
        return _back(wa.relative_position_index)


def ops(reference=None):
    return _Ops(reference)


def models(reference=None):
    return _Models(reference)
