"""saver/h5_saver.py of the reference (:38-298) restated: name-based loading with the fuzzy weight-name match, through the .npz
container (h5py is absent here) -- round trips, renamed weights, mismatches, get_backbone(weights_path=...)."""
import numpy as np
import pytest
import torch

from iseg_amd import nn
from iseg_amd.saver import (compute_string_similarity, load_h5_weight_by_name, open_weights, save_weights, search_weights, write_npz)


@pytest.fixture(autouse=True)
def cpu_device():
    prev = nn.device()
    nn.set_device("cpu")
    yield
    nn.set_device(prev)


def _backbone(seed):
    from iseg_amd.backbones.feature_extractor import get_backbone

    nn.set_seed(seed)
    return get_backbone("convnext_tiny", image_shape=(1, 64, 64, 3), return_endpoints=True)


def _state(model):
    d = {p.iseg_name: p.detach().clone() for p in model.parameters()}
    d.update({b.iseg_name: b.detach().clone() for b in model.buffers() if getattr(b, "iseg_name", None)})
    return d


def test_similarity_and_search_follow_the_reference():
    assert compute_string_similarity("a/b/c:0", "a/b/c:0") == 1.0
    assert compute_string_similarity("stages/0/dwconv/kernel:0", "model.stages.0.dwconv.kernel:0") == pytest.approx(5 / 6)
    assert compute_string_similarity("x/kernel:0", "y/bias:0") == pytest.approx(1 / 3)      # only the ':0' suffix is shared
    kv = {"m/conv/kernel:0": np.zeros((3, 3, 4, 8)), "m/conv/bias:0": np.ones(8), "m/other/kernel:0": np.full((3, 3, 4, 8), 2.0)}
    assert search_weights("m/conv/bias:0", kv, (8,)) is kv["m/conv/bias:0"]                      # exact name
    assert search_weights("net/conv/kernel:0", kv, (3, 3, 4, 8)) is kv["m/conv/kernel:0"]        # best shared components among same shape
    assert search_weights("net/conv/kernel:0", kv, (5,)) is None                                  # nothing of that shape
    assert search_weights("zzz", kv, None) is kv["m/conv/kernel:0"]                               # all scores 0: first in file order


def test_round_trip_restores_every_weight(tmp_path):
    a, b = _backbone(1), _backbone(2)
    sa = _state(a)
    assert any(not torch.equal(sa[k], v) for k, v in _state(b).items())
    path = save_weights(a, str(tmp_path / "convnext.npz"))
    n = load_h5_weight_by_name(b, path)
    sb = _state(b)
    assert n == len(sa) and all(torch.equal(sa[k], sb[k]) for k in sa)


class _Head(nn.Layer):
    """three direct layers, one file group each -- the granularity of a functional Keras model"""

    def __init__(self):
        super().__init__(name="head")
        from iseg_amd.layers.model_builder import ConvNormAct

        self.a = ConvNormAct(16, 3, name="branch_a")
        self.b = ConvNormAct(16, 3, use_bias=True, name="branch_b")
        self.c = ConvNormAct(8, 1, name="fuse")

    def call(self, x, training=None):
        return self.c(self.a(x, training=training) + self.b(x, training=training), training=training)


def _head(seed):
    nn.set_seed(seed)
    m = _Head()
    with nn.dry_run_scope():
        m(torch.empty(1, 8, 8, 16))
    from tests.util_models import randomize_parameters

    randomize_parameters(m, seed)
    return m


def test_renamed_weights_still_find_their_place(tmp_path):
    """a file written by another Keras version: '.'-separated layer names (replace_slash), a model prefix in front of every weight name,
    another separator -- inside each layer group the fuzzy match puts every array where its shape and name components say.  (Like the
    reference's, the match is by component *presence*: it cannot tell two same-shaped weights of one group apart, e.g. gamma / beta of a
    norm -- which is why exact names are tried first; here 'gamma' / 'beta' / 'moving_mean' / 'moving_variance' stay in the names.)"""
    a, b = _head(3), _head(4)
    root = open_weights(save_weights(a, str(tmp_path / "a.npz")))
    layers = {}
    for lname in [str(s) for s in root.attrs["layer_names"]]:
        g = root[lname]
        layers[lname.replace("/", ".")] = {"seg_model." + w.replace("/", "."): g[w] for w in [str(s) for s in g.attrs["weight_names"]]}
    path = write_npz(str(tmp_path / "renamed.npz"), layers)
    load_h5_weight_by_name(b, path)
    sa, sb = _state(a), _state(b)
    assert len(sa) == 16 and all(torch.equal(sa[k], sb[k]) for k in sa)


def test_mismatches_raise_or_skip(tmp_path):
    a, b = _backbone(5), _backbone(6)
    root = open_weights(save_weights(a, str(tmp_path / "a.npz")))
    names = [str(s) for s in root.attrs["layer_names"]]
    layers = {n: {w: root[n][w] for w in [str(s) for s in root[n].attrs["weight_names"]]} for n in names}
    first = names[0]
    wn = list(layers[first].keys())
    dropped = dict(layers)
    dropped[first] = {w: layers[first][w] for w in wn[1:]}                    # one array missing
    p1 = write_npz(str(tmp_path / "count.npz"), dropped)
    with pytest.raises(ValueError, match="Weight count mismatch"):
        load_h5_weight_by_name(b, p1)
    with pytest.warns(UserWarning):
        load_h5_weight_by_name(b, p1, skip_mismatch=True)
    reshaped = dict(layers)
    reshaped[first] = dict(layers[first])
    reshaped[first][wn[0]] = np.zeros((1, 2, 3), np.float32)                  # nothing of the expected shape is left for one weight
    big = [w for w in wn if layers[first][w].shape == layers[first][wn[0]].shape]
    if len(big) == 1:
        p2 = write_npz(str(tmp_path / "shape.npz"), reshaped)
        with pytest.raises(ValueError, match="Shape mismatch"):
            load_h5_weight_by_name(b, p2)
    extra = dict(layers)
    extra["no_such_layer"] = {"no_such_layer/kernel:0": np.zeros(3, np.float32)}
    with pytest.warns(UserWarning, match="no layer with this name"):
        load_h5_weight_by_name(b, write_npz(str(tmp_path / "extra.npz"), extra))


def test_get_backbone_loads_weights_path_and_rejects_unknown_formats(tmp_path):
    from iseg_amd.backbones.feature_extractor import get_backbone

    a = _backbone(7)
    path = save_weights(a, str(tmp_path / "convnext_tiny.h5.npz"))
    nn.set_seed(8)
    b = get_backbone("convnext_tiny", image_shape=(1, 64, 64, 3), return_endpoints=True, weights_path=path)
    sa, sb = _state(a), _state(b)
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    topo = save_weights(a, str(tmp_path / "convnext_tiny.topology.h5.npz"))
    nn.set_seed(9)
    c = get_backbone("convnext_tiny", image_shape=(1, 64, 64, 3), return_endpoints=True, weights_path=topo)
    assert all(torch.equal(sa[k], v) for k, v in _state(c).items())
    with pytest.raises(ValueError, match="not supported"):
        get_backbone("convnext_tiny", image_shape=(1, 64, 64, 3), weights_path=str(tmp_path / "weights.bin"))
    with pytest.raises(ImportError, match="h5py"):
        (tmp_path / "real.h5").write_bytes(b"\x89HDF\r\n\x1a\n")
        get_backbone("convnext_tiny", image_shape=(1, 64, 64, 3), weights_path=str(tmp_path / "real.h5"))


def test_resnet_takes_the_strict_by_name_route(tmp_path):
    """backbones/feature_extractor.py:174-176: for a ResNet the stored arrays go to the layer of the same name BY POSITION (Keras'
    load_weights_from_hdf5_group_by_name), so BatchNormalization's four same-shaped vectors cannot be swapped by a fuzzy name match -- also
    when the file names them differently; a count mismatch raises"""
    from iseg_amd.backbones.feature_extractor import get_backbone
    from iseg_amd.utils.keras_ops import get_all_layers

    nn.set_seed(21)
    a = get_backbone("resnet50", image_shape=(1, 64, 64, 3), return_endpoints=True)
    from tests.util_models import randomize_parameters

    randomize_parameters(a, 21)
    # one group per LEAF layer that owns weights (what save_model_to_h5(get_all_layers(model)) of the reference writes), weights renamed so
    # that only their order identifies them
    layers = {}
    for layer in get_all_layers(a):
        own = [(p.iseg_name, p) for p in layer.parameters(recurse=False)] + [(b.iseg_name, b) for b in layer.buffers(recurse=False)
                                                                               if getattr(b, "iseg_name", None)]
        if own:
            layers[layer.name] = {f"w{i}:0": t.detach().cpu().numpy() for i, (_, t) in enumerate(own)}
    path = write_npz(str(tmp_path / "resnet50.h5.npz"), layers)
    nn.set_seed(22)
    b = get_backbone("resnet50", image_shape=(1, 64, 64, 3), return_endpoints=True, weights_path=path)
    sa, sb = _state(a), _state(b)
    assert len(sa) > 200 and all(torch.equal(sa[k], sb[k]) for k in sa)
    first = next(iter(layers))
    broken = dict(layers)
    broken[first] = dict(list(layers[first].items())[:-1]) if len(layers[first]) > 1 else {**layers[first], "extra:0": np.zeros(3, np.float32)}
    with pytest.raises(ValueError, match="Weight count mismatch"):
        get_backbone("resnet50", image_shape=(1, 64, 64, 3), return_endpoints=True, weights_path=write_npz(str(tmp_path / "broken.h5.npz"), broken))


def test_eva_position_embedding_is_resampled_from_the_pretrain_grid_on_its_first_assignment(tmp_path):
    """round-5 advisor: the reference's Eva.build wraps pos_embed.assign so the first value assigned -- the pretrained table on the grid
    pretrain_img_size // patch -- is resampled bicubically to the build grid, prefix token passed through (backbones/eva/eva.py:131-165,
    utils/common.py:206-262); e.g. the 24 x 24 table of a 336-pixel EVA02 checkpoint poured into a 448-pixel (32 x 32) build"""
    from iseg_amd.backbones.eva.eva import Eva
    from iseg_amd.utils.bicubic import bicubic_matrix

    def make(size):
        nn.set_seed(3)
        m = Eva(pretrain_img_size=56, pretrain_patch_size=14, patch_size=14, embed_filters=16, depth=1, num_heads=2, name="eva_small")
        with nn.dry_run_scope():
            m(torch.empty(1, size, size, 3))
        return m

    src, dst = make(56), make(84)      # grids 4 x 4 (= the pretrain grid) and 6 x 6
    assert tuple(src.position_embedding.shape) == (1, 17, 16) and tuple(dst.position_embedding.shape) == (1, 37, 16)
    table = torch.randn(1, 17, 16)
    with torch.no_grad():
        src.position_embedding.copy_(table)

    class Holder(nn.Layer):
        def __init__(self, inner):
            super().__init__(name="holder")
            self.inner = inner
            self.built = True

    path = save_weights(Holder(src), str(tmp_path / "eva.npz"))
    n = load_h5_weight_by_name(Holder(dst), path)
    assert n > 0
    want = np.einsum("oh,hwc->owc", bicubic_matrix(6, 4), table[0, 1:].reshape(4, 4, 16).numpy())
    want = np.einsum("pw,owc->opc", bicubic_matrix(6, 4), want).reshape(36, 16)
    got = dst.position_embedding.detach().numpy()
    assert np.array_equal(got[0, 0], table[0, 0].numpy())                  # the class-token slot passes through
    assert np.abs(got[0, 1:] - want).max() < 1e-6
    # a second assignment (a checkpoint of THIS model, already on the build grid) is taken as it comes
    again = torch.randn(1, 37, 16).numpy()
    assert np.array_equal(dst._resample_on_first_assign(again), again)
    # a table on neither grid is an error, not a silent reshape
    fresh = make(84)
    with pytest.raises(ValueError):
        fresh._resample_on_first_assign(np.zeros((1, 1 + 25, 16), np.float32))
