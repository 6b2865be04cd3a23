"""Weight containers for the name-based loader (saver/h5_saver.py).

The reference reads Keras HDF5 weight files through h5py (saver/h5_saver.py:38-46, utils/keras_ops.py:107-127).  h5py is not part of
this image, so the same *structure* -- a root group with the attribute `layer_names`, one sub-group per layer with the attribute
`weight_names` and one array per weight, optionally `top_level_model_weights` -- is also stored as a flat `.npz`:

    "@layer_names"                  array of str
    "@keras_version", "@backend"    0-d str arrays (optional)
    "@weight_names/<layer>"         array of str, the stored weight names of the layer, in order
    "<layer>::<weight name>"        the values

`tools/h5_to_npz.py` turns a Keras `.h5` / `.weights.h5` into that file wherever h5py exists; `open_weights` reads either kind (the
`.h5` branch needs h5py and says so).  `save_weights` writes the `.npz` form of a model built here, so checkpoints of this package and
converted reference weights go through one loader."""
import os

import numpy as np


class WeightGroup:
    """minimal h5py.Group look-alike over plain dicts: .attrs, `name in g`, g[name] -> WeightGroup | ndarray"""

    def __init__(self, attrs=None, members=None):
        self.attrs = dict(attrs or {})
        self.members = dict(members or {})

    def __contains__(self, k):
        return k in self.members

    def __getitem__(self, k):
        return self.members[k]

    def keys(self):
        return self.members.keys()


def _attr_list(group, name):
    """keras' load_attributes_from_hdf5_group: an attribute too large for one HDF5 header is split into name0, name1, ..."""
    def text(v):
        return v.decode("utf8") if hasattr(v, "decode") else str(v)

    if name in group.attrs:
        return [text(v) for v in np.asarray(group.attrs[name]).reshape(-1)]
    out, i = [], 0
    while f"{name}{i}" in group.attrs:
        out += [text(v) for v in np.asarray(group.attrs[f"{name}{i}"]).reshape(-1)]
        i += 1
    return out


def layer_names_of(group):
    return _attr_list(group, "layer_names")


def weight_names_of(group):
    return _attr_list(group, "weight_names")


def _from_npz(path):
    z = np.load(path, allow_pickle=False)
    root = WeightGroup()
    for key in ("keras_version", "backend"):
        if f"@{key}" in z.files:
            root.attrs[key] = str(z[f"@{key}"])
    names = [str(s) for s in z["@layer_names"]] if "@layer_names" in z.files else []
    root.attrs["layer_names"] = np.asarray(names)
    groups = list(names)
    if "@weight_names/top_level_model_weights" in z.files:
        groups.append("top_level_model_weights")
    for layer in groups:
        wn = [str(s) for s in z[f"@weight_names/{layer}"]] if f"@weight_names/{layer}" in z.files else []
        g = WeightGroup({"weight_names": np.asarray(wn)})
        for w in wn:
            g.members[w] = z[f"{layer}::{w}"]
        root.members[layer] = g
    return root


def _from_h5(path):
    try:
        import h5py
    except ImportError as e:
        raise ImportError(f"{path}: reading Keras HDF5 weights needs h5py, which this environment does not have. Convert the file where "
                          "h5py is installed with `python tools/h5_to_npz.py weights.h5 weights.npz` and pass the .npz instead.") from e

    def walk_weights(g):
        grp = WeightGroup({k: np.asarray(v) for k, v in g.attrs.items()})
        for w in _attr_list(grp, "weight_names"):
            grp.members[w] = np.asarray(g[w])
        return grp

    with h5py.File(path, "r") as f:
        if "layer_names" not in f.attrs and "model_weights" in f:      # a full-model file: the weights live one level down
            f = f["model_weights"]
        root = WeightGroup({k: np.asarray(v) for k, v in f.attrs.items()})
        for layer in layer_names_of(root):
            root.members[layer] = walk_weights(f[layer])
        if "top_level_model_weights" in f:
            root.members["top_level_model_weights"] = walk_weights(f["top_level_model_weights"])
        return root


def open_weights(path):
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    if path.endswith(".npz"):
        return _from_npz(path)
    if path.endswith(".h5") or path.endswith(".hdf5"):
        return _from_h5(path)
    raise ValueError(f"Weights {path} not supported")


def write_npz(path, layers, top_level=None, keras_version="iseg_amd", backend="hip"):
    """layers: ordered {layer name: ordered {stored weight name: ndarray}}"""
    out = {"@layer_names": np.asarray(list(layers.keys()), dtype=str), "@keras_version": np.asarray(keras_version),
           "@backend": np.asarray(backend)}
    groups = dict(layers)
    if top_level:
        groups["top_level_model_weights"] = top_level
    for layer, ws in groups.items():
        out[f"@weight_names/{layer}"] = np.asarray(list(ws.keys()), dtype=str)
        for w, v in ws.items():
            out[f"{layer}::{w}"] = np.asarray(v)
    tmp = path + ".tmp.npz"
    np.savez(tmp, **out)
    os.replace(tmp, path)
    return path
