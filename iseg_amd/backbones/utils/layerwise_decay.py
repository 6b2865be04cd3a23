"""backbones/utils/layerwise_decay.py of the reference (:12-56)."""
from ...utils.train_utils import set_weights_lr_multiplier


def decay_layers_lr(layers=[], weights=[], rate=0.99):
    num_layers = len(layers)
    for i in range(num_layers):
        layer = layers[i]
        current_rate = rate ** (num_layers - i - 2)
        if isinstance(layer, tuple):
            layer = list(layer)
        if not isinstance(layer, list):
            layer = [layer]
        for sub_layer in layer:
            for v in sub_layer.trainable_weights:
                mult = current_rate
                if hasattr(v, "lr_multiplier"):
                    mult *= v.lr_multiplier
                set_weights_lr_multiplier(v, lr_multiplier=mult)
    current_rate = rate ** (num_layers - 1)
    for weight in weights:
        mult = getattr(weight, "lr_multiplier", 1.0)
        set_weights_lr_multiplier(weight, lr_multiplier=mult * current_rate)
