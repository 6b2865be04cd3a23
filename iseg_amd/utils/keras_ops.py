"""utils/keras_ops.py of the reference: get_all_layers (:20-37), set_weight_decay (:40-62), set_bn_momentum / set_bn_epsilon
(:80-99), capture_func."""
from ..layers.base_layers import BatchNormalization
from ..nn import Layer


def get_all_layers(model):
    return [m for m in model.modules() if isinstance(m, Layer)]


get_all_layers_v2 = get_all_layers


def set_weight_decay(model, weight_decay=0.0001, decay_norm_vars=False):
    """keras l2 kernel regularizers: loss += wd * sum(w^2)  ->  the SGD kernel adds 2*wd*w to the gradient"""
    for layer in get_all_layers(model):
        for attr in ("kernel", "depthwise_kernel"):
            k = getattr(layer, attr, None)
            if k is not None and hasattr(k, "requires_grad"):
                k.l2_regularizer = float(weight_decay)
        if decay_norm_vars:
            for attr in ("beta", "gamma"):
                v = getattr(layer, attr, None)
                if v is not None and hasattr(v, "requires_grad"):
                    v.l2_regularizer = float(weight_decay)


def set_bn_momentum(model, momentum=0.99):
    for layer in get_all_layers(model):
        if isinstance(layer, BatchNormalization):
            layer.momentum = momentum


def set_bn_epsilon(model, epsilon=1e-3):
    for layer in get_all_layers(model):
        if isinstance(layer, BatchNormalization):
            layer.epsilon = epsilon


def load_h5_weight(model, path, skip_mismatch=False, use_v2_behavior=False, by_name=True):
    """utils/keras_ops.py:107-127: Keras HDF5 weights (or their .npz conversion, saver/weights_file.py) into `model`, by layer name or
    -- by_name=False, the `.topology.h5` files of backbones/feature_extractor.py:167-169 -- by layer order"""
    from ..saver import load_weights_from_group_by_name, load_weights_from_group_by_name_strict, load_weights_from_group_topological, open_weights

    f = open_weights(path)
    if by_name:
        if use_v2_behavior:      # (:115-116) the v2 loader = the fuzzy per-layer search of saver/h5_saver.py
            return load_weights_from_group_by_name(f, model, skip_mismatch=skip_mismatch)
        return load_weights_from_group_by_name_strict(f, get_all_layers(model), skip_mismatch=skip_mismatch)
    return load_weights_from_group_topological(f, model)


def capture_func(obj, name):
    fn = getattr(obj, name, None)
    return fn if callable(fn) else None
