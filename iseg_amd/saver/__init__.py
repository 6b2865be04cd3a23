"""saver/ of the reference: name-based weight loading for pretrained backbones (h5_saver.py)."""
from .h5_saver import (compute_string_similarity, load_h5_weight_by_name, load_weights_by_name, load_weights_from_group_by_name, load_weights_from_group_by_name_strict,  # noqa: F401
                       load_weights_from_group_topological, save_weights, search_weights)
from .weights_file import open_weights, write_npz  # noqa: F401
