"""utils/train_utils.py of the reference: no-weight-decay name list (:8-37), exclude_no_weight_decay_layers_in_optimizer
(:40-72), set_weights_lr_multiplier (:75-87)."""
from ..layers.base_layers import BatchNormalization, LayerNormalization
from .keras_ops import get_all_layers_v2


def get_no_weight_decay_layers_names_from_model(model):
    excluded_name_list = ["bias", "relative_position_bias_table", "pos", "patch_embed", "class_token", "logits"]
    for layer in get_all_layers_v2(model):
        if isinstance(layer, (LayerNormalization, BatchNormalization)):
            excluded_name_list.append(layer.name)
        else:
            layer_type_name = layer.__class__.__name__.lower()
            layer_name = layer.name.lower()
            if "norm" in layer_type_name:
                excluded_name_list.append(layer.name)
            if "logits" in layer_name:
                excluded_name_list.append(layer.name)
    return excluded_name_list


def exclude_no_weight_decay_layers_in_optimizer(optimizer, model, excluded_name_list=None, print_excluded_list=True):
    if excluded_name_list is None:
        excluded_name_list = get_no_weight_decay_layers_names_from_model(model=model)
    if isinstance(optimizer, list):
        for opt in optimizer:
            exclude_no_weight_decay_layers_in_optimizer(opt, model, excluded_name_list, print_excluded_list)
        return
    fn = getattr(optimizer, "exclude_from_weight_decay", None)
    if fn is None or not callable(fn):
        return
    if print_excluded_list:
        print(f"Excluded vars for weight decay: {len(excluded_name_list)} name patterns")
    fn(var_names=excluded_name_list)


def set_weights_lr_multiplier(var_list, lr_multiplier=1.0):
    if isinstance(var_list, tuple):
        var_list = list(var_list)
    if not isinstance(var_list, list):
        var_list = [var_list]
    for v in var_list:
        v.lr_multiplier = lr_multiplier
