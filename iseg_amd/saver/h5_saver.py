"""saver/h5_saver.py of the reference (:38-298): name-based weight loading -- layers are found by name, and inside a layer every model
weight takes the stored weight with the same name or, failing that, the stored weight of the same shape whose name shares the most
path components (`search_weights` / `compute_string_similarity`, :244-298).  Used by get_backbone(weights_path=...)
(backbones/feature_extractor.py:166-187) to pour pretrained backbones into a freshly built model."""
import re
import warnings

import numpy as np
import torch

from .. import nn
from ..nn import Layer
from ..utils.slash_utils import replace_slash
from .weights_file import layer_names_of, open_weights, weight_names_of, write_npz


def compute_string_similarity(a: str, b: str):
    """fraction of the path components of `a` (split at '/', '.', ':') that also occur in `b`  (:281-298)"""
    a_parts, b_parts = re.split(r"[\/\.\:]", a), re.split(r"[\/\.\:]", b)
    hits = sum(1 for pa in a_parts if pa != "" and pa in b_parts)
    return float(hits) / float(max(len(a_parts), len(b_parts)))


def search_weights(keyword, weights_kv_dict, expected_shape=None):
    """the stored value for model weight `keyword`: exact name, else the best-scoring name among the values of the expected shape;
    ties keep the first in file order (:244-278)"""
    if keyword in weights_kv_dict:
        return weights_kv_dict[keyword]
    best_key, best = None, None
    for key, value in weights_kv_dict.items():
        if expected_shape is not None and tuple(value.shape) != tuple(expected_shape):
            continue
        score = compute_string_similarity(keyword, key)
        if best is None or score > best:
            best_key, best = key, score
    return None if best_key is None else weights_kv_dict[best_key]


def load_subset_weights_kv_dict_from_hdf5_group(group):
    return {name: np.asarray(group[name]) for name in weight_names_of(group)}


def direct_layers(model):
    """keras' model.layers: the layers a model holds directly (lists of layers are flattened)"""
    out = []

    def visit(module):
        for child in module.children():
            if isinstance(child, Layer):
                out.append(child)
            else:      # ModuleList / Sequential containers
                visit(child)

    visit(model)
    return out


def layer_weights(layer):
    """[(stored-style name, tensor)] of a layer, nested layers included: trainable first, then the non-trainable state (keras order)"""
    ws = [(p.iseg_name + ":0", p) for p in layer.parameters()]
    ws += [(b.iseg_name + ":0", b) for b in layer.buffers() if getattr(b, "iseg_name", None)]
    return ws


def _norm(name):
    return name.replace("/", ".")


def _assign(pairs):
    with torch.no_grad():
        for target, value in pairs:
            hook = getattr(target, "iseg_assign_hook", None)      # e.g. Eva's position embedding: resampled to the build grid on its first assignment
            if hook is not None:
                value = hook(np.asarray(value))
            t = torch.as_tensor(np.asarray(value), dtype=torch.float32).reshape(tuple(target.shape)).to(target.device)
            (target.data if isinstance(target, torch.nn.Parameter) else target).copy_(t)
            shadow = getattr(target, "iseg_compute", None)
            if shadow is not None:
                shadow.copy_(t.to(shadow.dtype))
                nn.weights_changed()


def load_weights_from_group_by_name(f, model, skip_mismatch=False):
    """load_weights_from_hdf5_group_by_name_v3 (:49-229)"""
    index = {}
    for layer in direct_layers(model):
        if layer.name:
            index.setdefault(_norm(layer.name), []).append(layer)
    pairs, loaded_layers = [], 0
    for k, name in enumerate(layer_names_of(f)):
        kv = load_subset_weights_kv_dict_from_hdf5_group(f[name])
        name = replace_slash(name)
        layer_list = index.get(_norm(name), [])
        if not layer_list:
            warnings.warn(f'Skipping loading weights for layer "{name}" as no layer with this name exists in the model.')
        for layer in layer_list:
            symbolic = layer_weights(layer)
            if len(kv) != len(symbolic):
                msg = (f"Weight count mismatch for layer #{k} (named {layer.name}). Layer expects {len(symbolic)} weight(s). "
                       f"Received {len(kv)} saved weight(s)")
                if skip_mismatch:
                    warnings.warn("Skipping: " + msg)
                    continue
                raise ValueError(msg)
            loaded_layers += 1
            for wname, target in symbolic:
                # (a weight with an assign hook -- Eva's position embedding -- is taken at its stored shape: the hook resamples it)
                stored = search_weights(wname, kv, None if getattr(target, "iseg_assign_hook", None) is not None else tuple(target.shape))
                if stored is None:
                    msg = f"Shape mismatch in layer #{k} (named {layer.name}) for weight {wname}. Weight expects shape {tuple(target.shape)}."
                    if skip_mismatch:
                        warnings.warn("Skipping: " + msg)
                        continue
                    raise ValueError(msg)
                pairs.append((target, stored))
    if "top_level_model_weights" in f:
        own = [(p.iseg_name + ":0", p) for p in model.parameters(recurse=False)]
        own += [(b.iseg_name + ":0", b) for b in model.buffers(recurse=False) if getattr(b, "iseg_name", None)]
        kv = load_subset_weights_kv_dict_from_hdf5_group(f["top_level_model_weights"])
        if len(kv) != len(own):
            msg = (f"Weight count mismatch for top-level weights of model. Model expects {len(own)} top-level weight(s). "
                   f"Received {len(kv)} saved top-level weight(s)")
            if not skip_mismatch:
                raise ValueError(msg)
            warnings.warn("Skipping: " + msg)
        else:
            for wname, target in own:
                # a position embedding of another resolution is taken as stored and resized by the model (:186-190)
                stored = search_weights(wname, kv, tuple(target.shape) if "pos_embed" not in wname else None)
                if stored is None:
                    msg = f"Shape mismatch in model for top-level weight {wname}. Weight expects shape {tuple(target.shape)}."
                    if not skip_mismatch:
                        raise ValueError(msg)
                    warnings.warn("Skipping: " + msg)
                elif tuple(np.asarray(stored).shape) == tuple(target.shape):
                    pairs.append((target, stored))
                else:
                    warnings.warn(f"{wname}: stored shape {tuple(np.asarray(stored).shape)} differs from {tuple(target.shape)}; left as built")
    _assign(pairs)
    return len(pairs)


def load_weights_from_group_by_name_strict(f, layers, skip_mismatch=False):
    """keras load_weights_from_hdf5_group_by_name (utils/hdf5_utils.py:230-383), the route the reference keeps for ResNets
    (backbones/feature_extractor.py:174-176 -> utils/keras_ops.py:107-122 with layers = get_all_layers(model)): a stored group goes to every
    layer of that NAME, its arrays to the layer's weights BY POSITION -- count and shapes must agree.  No fuzzy search: two same-shaped
    weights of one layer (BatchNormalization's gamma / beta / moving_mean / moving_variance) cannot change places."""
    index = {}
    for layer in layers:
        if layer.name:
            index.setdefault(layer.name, []).append(layer)
    pairs = []
    for k, name in enumerate(layer_names_of(f)):
        values = list(load_subset_weights_kv_dict_from_hdf5_group(f[name]).values())
        for layer in index.get(name, []):
            symbolic = layer_weights(layer)
            if len(values) != len(symbolic):
                msg = (f"Weight count mismatch for layer #{k} (named {layer.name}). Layer expects {len(symbolic)} weight(s). "
                       f"Received {len(values)} saved weight(s)")
                if skip_mismatch:
                    warnings.warn("Skipping loading weights for layer #{} (named {}) due to mismatch in number of weights.".format(k, layer.name))
                    continue
                raise ValueError(msg)
            ok = True
            for (wname, target), v in zip(symbolic, values):
                if tuple(np.asarray(v).shape) != tuple(target.shape):
                    msg = (f"Shape mismatch in layer #{k} (named {layer.name}) for weight {wname}. Weight expects shape {tuple(target.shape)}. "
                           f"Received saved weight with shape {tuple(np.asarray(v).shape)}")
                    if not skip_mismatch:
                        raise ValueError(msg)
                    warnings.warn("Skipping: " + msg)
                    ok = False
            if ok:
                pairs.extend((target, v) for (_, target), v in zip(symbolic, values))
    _assign(pairs)
    return len(pairs)


def load_h5_weight_by_name(model, path, skip_mismatch=False):
    """(:38-46) `path`: a Keras .h5 (needs h5py) or its .npz conversion (tools/h5_to_npz.py) or a file written by save_weights"""
    return load_weights_from_group_by_name(open_weights(path), model, skip_mismatch=skip_mismatch)


load_weights_by_name = load_h5_weight_by_name


def load_weights_from_group_topological(f, model):
    """keras load_weights_from_hdf5_group (utils/hdf5_utils.py:386-477): the k-th stored layer that has weights goes to the k-th model
    layer that has weights, weights in order"""
    model_layers = [l for l in direct_layers(model) if layer_weights(l)]
    stored = [(n, load_subset_weights_kv_dict_from_hdf5_group(f[n])) for n in layer_names_of(f)]
    stored = [(n, kv) for n, kv in stored if kv]
    if len(stored) != len(model_layers):
        raise ValueError(f"Layer count mismatch when loading weights from file. Model expected {len(model_layers)} layers, found "
                         f"{len(stored)} saved layers.")
    pairs = []
    for (n, kv), layer in zip(stored, model_layers):
        symbolic = layer_weights(layer)
        values = list(kv.values())
        if len(values) != len(symbolic):
            raise ValueError(f"Weight count mismatch for layer (named {layer.name} in the current model, {n} in the save file). Layer "
                             f"expects {len(symbolic)} weight(s). Received {len(values)} saved weight(s)")
        for (wname, target), v in zip(symbolic, values):
            if tuple(np.asarray(v).shape) != tuple(target.shape):
                raise ValueError(f"Shape mismatch for weight {wname}: expects {tuple(target.shape)}, received {tuple(np.asarray(v).shape)}")
            pairs.append((target, v))
    _assign(pairs)
    return len(pairs)


def save_weights(model, path):
    """the model's weights in the layer-group layout this loader reads (.npz); returns the path"""
    layers = {}
    for layer in direct_layers(model):
        ws = layer_weights(layer)
        if ws:
            layers.setdefault(layer.name, {}).update({n: t.detach().cpu().float().numpy() for n, t in ws})
    top = {p.iseg_name + ":0": p.detach().cpu().float().numpy() for p in model.parameters(recurse=False)}
    return write_npz(path, layers, top or None)
