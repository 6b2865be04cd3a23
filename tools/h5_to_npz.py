#!/usr/bin/env python3
"""Convert a Keras HDF5 weight file (the pretrained backbones of the reference's backbones/README.md, `model.save_weights("x.h5")`, or a
full-model .h5) into the flat .npz that iseg_amd.saver reads (layout: iseg_amd/saver/weights_file.py).  Needs h5py, so run it wherever
the reference's own environment exists; the result is plain numpy and travels anywhere.
usage: python3 tools/h5_to_npz.py weights.h5 [weights.npz]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    if len(sys.argv) < 2:
        print(__doc__)
        return 2
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) > 2 else src + ".npz"
    from iseg_amd.saver.weights_file import layer_names_of, open_weights, weight_names_of, write_npz

    root = open_weights(src)      # raises with an explanation when h5py is missing
    layers = {}
    for name in layer_names_of(root):
        g = root[name]
        layers[name] = {w: g[w] for w in weight_names_of(g)}
    top = None
    if "top_level_model_weights" in root:
        g = root["top_level_model_weights"]
        top = {w: g[w] for w in weight_names_of(g)}
    write_npz(dst, layers, top, keras_version=str(root.attrs.get("keras_version", "")), backend=str(root.attrs.get("backend", "")))
    n = sum(len(v) for v in layers.values()) + (len(top) if top else 0)
    print(f"{src} -> {dst}: {len(layers)} layers, {n} arrays")
    return 0


if __name__ == "__main__":
    sys.exit(main())
