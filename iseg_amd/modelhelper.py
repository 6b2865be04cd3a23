"""modelhelper.py of the reference: model_common_setup (:22-56), ModelHelper (:59-264; timestamped checkpoints, keep newest N,
optimizer state is not checkpointed -- LR step is rebuilt from initial_epoch * epoch_steps)."""
import glob
import os
import time

import torch

from .param_store import ParamStore
from .utils.keras_ops import set_bn_epsilon, set_bn_momentum, set_weight_decay

POSTFIX = "ckpt.weights.pt"


def model_common_setup(model, restore_checkpoint=True, checkpoint_dir=None, max_checkpoints_to_keep=1, weight_decay=None,
                       decay_norm_vars=False, bn_epsilon=None, bn_momentum=None, backbone_bn_momentum=None,
                       inference_sliding_window_size=None):
    model.inference_sliding_window_size = inference_sliding_window_size
    if hasattr(model, "build_with_dummy"):
        model.build_with_dummy()
    model_helper = ModelHelper(model, checkpoint_dir, max_checkpoints_to_keep)
    if restore_checkpoint:
        model_helper.restore_checkpoint()
    if weight_decay is not None and weight_decay > 0:
        set_weight_decay(model_helper.model, weight_decay, decay_norm_vars)
    if bn_epsilon is not None:
        set_bn_epsilon(model_helper.model, bn_epsilon)
    if bn_momentum is not None:
        set_bn_momentum(model_helper.model, bn_momentum)
    if backbone_bn_momentum is not None and hasattr(model_helper.model, "backbone"):
        set_bn_momentum(model_helper.model.backbone, backbone_bn_momentum)
    return model_helper


class ModelHelper:
    def __init__(self, model, checkpoint_dir, max_to_keep=20, force_use_keras2=False):
        self.model = model
        self.checkpoint_dir = checkpoint_dir
        self.max_to_keep = max_to_keep
        self.__optimizer = None
        if getattr(model, "_iseg_store", None) is None and any(True for _ in model.parameters()):
            model._iseg_store = ParamStore(list(model.parameters()))

    def set_optimizer(self, optimizer):
        self.__optimizer = optimizer

    @property
    def optimizer(self):
        if self.__optimizer is None:
            raise ValueError("The optimizer is None")
        return self.__optimizer

    def _named_tensors(self):
        out = {}
        for p in self.model.parameters():
            out[p.iseg_name] = p.data
        for b in self.model.buffers():
            out[getattr(b, "iseg_name", None) or f"buffer_{len(out)}"] = b
        return out

    def list_checkpoints(self):
        if self.checkpoint_dir is None or not os.path.isdir(self.checkpoint_dir):
            return []
        return sorted(glob.glob(os.path.join(self.checkpoint_dir, f"id-*.{POSTFIX}")))

    def save_checkpoint(self):
        if self.checkpoint_dir is None:
            return None
        from . import dist

        if dist.rank() != 0:
            return None
        os.makedirs(self.checkpoint_dir, exist_ok=True)
        # the reference's names have one-second resolution (modelhelper.py:201-213); two saves inside one second must not
        # overwrite each other, so a per-helper counter follows the time stamp (lexicographic order = save order)
        self._save_count = getattr(self, "_save_count", 0) + 1
        path = os.path.join(self.checkpoint_dir, f"id-{time.strftime('%Y%m%d-%H%M%S')}-{self._save_count:06d}.{POSTFIX}")
        while os.path.exists(path):
            self._save_count += 1
            path = os.path.join(self.checkpoint_dir, f"id-{time.strftime('%Y%m%d-%H%M%S')}-{self._save_count:06d}.{POSTFIX}")
        torch.save({k: v.detach().cpu() for k, v in self._named_tensors().items()}, path)
        ckpts = self.list_checkpoints()
        for old in ckpts[:-self.max_to_keep] if self.max_to_keep > 0 else []:
            os.remove(old)
        return path

    def restore_checkpoint(self, skip_mismatch=False):
        """newest checkpoint -> model.  Every variable of the model must be found with its shape (and every stored tensor must
        have a home) unless skip_mismatch=True, in which case the unmatched names are reported and left untouched."""
        ckpts = self.list_checkpoints()
        if not ckpts:
            return None
        state = torch.load(ckpts[-1], map_location="cpu")
        mine = self._named_tensors()
        missing = [k for k in mine if k not in state]
        unexpected = [k for k in state if k not in mine]
        mismatched = [k for k in state if k in mine and tuple(mine[k].shape) != tuple(state[k].shape)]
        if missing or unexpected or mismatched:
            msg = (f"checkpoint {ckpts[-1]}: {len(missing)} model variables not in the file {missing[:5]}, {len(unexpected)} stored "
                   f"tensors without a variable {unexpected[:5]}, {len(mismatched)} shape mismatches {mismatched[:5]}")
            if not skip_mismatch:
                raise ValueError(msg + " (pass skip_mismatch=True to load the rest)")
            print("WARNING: " + msg)
        for k, v in state.items():
            if k in mine and tuple(mine[k].shape) == tuple(v.shape):
                mine[k].copy_(v)
        store = getattr(self.model, "_iseg_store", None)
        if store is not None:
            store.sync_shadow()
        return ckpts[-1]
