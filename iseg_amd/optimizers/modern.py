"""optimizers/modern/{adamw,sgd}.py of the reference: AdamW_EXT.update_step/_clip_gradients (adamw.py:13-74), SGD_EXT.update_step
(sgd.py:12-51), on top of Keras' optimizer base (decoupled weight decay with `exclude_from_weight_decay(var_names)`,
`iterations`, `clipnorm` / `global_clipnorm` / `clipvalue`).  One fused kernel launch over the flat parameter buffer (csrc/optim.hip),
preceded by the fixed-order norm reduction when a norm clip is configured."""
import math
import re

import torch

from .. import _hip
from .. import kernels as K
from .. import nn
from ..param_store import ParamStore


class _FlatOptimizer:
    _scrub_nan = False

    def __init__(self, learning_rate, clipnorm=None, clipvalue=None, global_clipnorm=None):
        self.learning_rate = learning_rate
        if sum(v is not None for v in (clipnorm, clipvalue, global_clipnorm)) > 1:      # Keras' base optimizer refuses the same
            raise ValueError(f"At most one of `clipnorm`, `clipvalue` and `global_clipnorm` can be set. Received: clipnorm={clipnorm}, "
                             f"clipvalue={clipvalue}, global_clipnorm={global_clipnorm}.")
        self.clipnorm, self.clipvalue, self.global_clipnorm = clipnorm, clipvalue, global_clipnorm
        self.iterations = 0
        self._exclude = []
        self.store = None
        self.grad_scale = 1.0
        self._owned = None      # None: every variable of the store; else the names this optimizer updates (MultiOptimizer)

    def restrict_to(self, var_names):
        """only the variables with these names are updated by this optimizer (optimizers/multi_optimizer.py routes (grad, var) pairs by
        name, :42-55): the others keep learning-rate multiplier 0 and no decay in this optimizer's tables, i.e. the step leaves them as they are"""
        self._owned = None if var_names is None else set(var_names)
        if self.store is not None:
            self._build_tables()

    def _owns(self, p):
        """this optimizer updates p: routed here (MultiOptimizer).  Frozen variables (Keras `layer.trainable = False`) are `requires_grad = False`
        parameters: the tables below give them learning-rate multiplier 0 and no decay"""
        return self._owned is None or getattr(p, "iseg_name", None) in self._owned

    # Keras API
    def exclude_from_weight_decay(self, var_list=None, var_names=None):
        self._exclude = list(var_names or [])
        self._exclude_vars = [id(v) for v in (var_list or [])]
        if self.store is not None:
            self._build_tables()

    def _use_weight_decay(self, p):
        if id(p) in getattr(self, "_exclude_vars", []):
            return False
        name = getattr(p, "iseg_name", "")
        return not any(re.search(n, name) is not None for n in self._exclude)

    def current_lr(self):
        lr = self.learning_rate
        return float(lr(self.iterations)) if callable(lr) else float(lr)

    def build(self, store: ParamStore):
        self.store = store
        dev = store.device
        # step-dependent scalars travel through a ring of pinned-host -> device slots.  The async copy of step t must not be overwritten
        # on the host before the GPU has consumed it: every slot carries the event recorded behind its last copy and is waited for
        # before it is rewritten (free while the host is less than a ring ahead of the stream)
        self._ring = 64
        self._hp_dev = torch.zeros(self._ring, 4, dtype=torch.float32, device=dev)
        self._hp_host = torch.zeros(self._ring, 4, dtype=torch.float32)
        self._hp_events = [None] * self._ring
        if dev.type == "cuda":
            self._hp_host = self._hp_host.pin_memory()
        self._slot = 0
        self.hp = self._hp_dev[0]
        self._hp_fixed = None        # graph mode (fixed_hp_slot): the one device slot a captured step kernel reads
        self._prepared = False       # prepare_step() already pushed this step's scalars (GraphedTrainStep does it outside the graph)
        seg_first = [o // 256 for (_, o, _) in store.segments] + [store.nblocks]
        self._seg_first_block = torch.tensor(seg_first, dtype=torch.int32, device=dev)
        self._norm_ws = None
        self._build_tables()

    def _build_tables(self):
        raise NotImplementedError

    def _push_hp(self, vals):
        slot = self._slot
        self._slot = (slot + 1) % self._ring
        ev = self._hp_events[slot]
        if ev is not None:
            ev.synchronize()
        for i, v in enumerate(vals):
            self._hp_host[slot, i] = v
        # eager: the device slot rotates with the host slot; graph mode: always the same device words (the captured kernel's argument),
        # filled by a stream-ordered copy from the rotating pinned slot in front of every replay
        self.hp = self._hp_dev[slot] if self._hp_fixed is None else self._hp_fixed
        self.hp.copy_(self._hp_host[slot], non_blocking=True)
        if self.hp.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            self._hp_events[slot] = ev

    def fixed_hp_slot(self, on=True):
        """graph mode: the step kernel reads its scalars from ONE device address (iseg_amd/graphs.py)"""
        # ONE slot for the optimizer's lifetime: every captured graph (one per input signature) holds this address, so a second capture must not
        # replace it -- the first graph would keep reading a freed slot that prepare_step() no longer fills.  Switching graph mode off only
        # parks it (`_hp_fixed_slot` stays alive for the graphs that still point at it).
        if on:
            if getattr(self, "_hp_fixed_slot", None) is None:
                self._hp_fixed_slot = torch.zeros(4, dtype=torch.float32, device=self.store.device)
            self._hp_fixed = self._hp_fixed_slot
        else:
            self._hp_fixed = None

    def _hp_values(self):
        raise NotImplementedError

    def prepare_step(self):
        """host half of apply_gradients(): this step's learning rate / bias correction / gradient scale / clip value on their way to the
        device.  apply_gradients() calls it itself unless a graph runner already did (outside the captured region)."""
        self._push_hp(self._hp_values())
        self._prepared = True

    def after_replayed_step(self):
        """host bookkeeping of a step whose kernels were replayed from a graph"""
        self.iterations += 1
        self._prepared = False
        nn.weights_changed()

    def _clip_tables(self, seg_l2=None):
        """(seg_sq, clipnorm, global_sq, global_clipnorm) for the step kernel; runs iseg_grad_sqnorm when a norm clip is configured"""
        cn = float(self.clipnorm) if self.clipnorm and self.clipnorm > 0 else 0.0
        gn = float(self.global_clipnorm) if self.global_clipnorm and self.global_clipnorm > 0 else 0.0
        if cn <= 0 and gn <= 0:
            return None, 0.0, None, 0.0
        st = self.store
        if self._norm_ws is None:
            self._norm_ws = (torch.empty(st.nblocks, dtype=torch.float32, device=st.device),
                             torch.empty(len(st.segments), dtype=torch.float32, device=st.device),
                             torch.empty(1, dtype=torch.float32, device=st.device))
        blk, seg_sq, tot = self._norm_ws
        _hip.call("iseg_grad_sqnorm", K.ptr(st.flat_g), K.ptr(st.flat_w), K.ptr(st.seg_of_block), K.ptr(self._seg_first_block), K.ptr(seg_l2),
                  K.ptr(self.hp), int(self._scrub_nan), K.ptr(blk), K.ptr(seg_sq), K.ptr(tot) if gn > 0 else None, st.nblocks,
                  len(st.segments), K.stream())
        return (seg_sq if cn > 0 else None), cn, (tot if gn > 0 else None), gn


class AdamW(_FlatOptimizer):
    _scrub_nan = True      # AdamW_EXT._clip_gradients replaces NaN gradients by 0 before clipping (adamw.py:63-74)

    def __init__(self, learning_rate=0.001, weight_decay=0.004, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False, clipnorm=None,
                 clipvalue=None, global_clipnorm=None, name="AdamW"):
        super().__init__(learning_rate, clipnorm, clipvalue, global_clipnorm)
        self.amsgrad = bool(amsgrad)
        self.weight_decay, self.beta_1, self.beta_2, self.epsilon = weight_decay, beta_1, beta_2, epsilon

    def _build_tables(self):
        st = self.store
        self.m = getattr(self, "m", None) if getattr(self, "m", None) is not None else torch.zeros_like(st.flat_w)
        self.v = getattr(self, "v", None) if getattr(self, "v", None) is not None else torch.zeros_like(st.flat_w)
        if self.amsgrad and getattr(self, "vhat", None) is None:
            self.vhat = torch.zeros_like(st.flat_w)
        lr_mult = [float(getattr(p, "lr_multiplier", 1.0)) if (p.requires_grad and self._owns(p)) else 0.0 for p in st.params]
        wd = [float(self.weight_decay or 0.0) if (p.requires_grad and self._owns(p) and self._use_weight_decay(p)) else 0.0 for p in st.params]
        self.seg_lr_mult = torch.tensor(lr_mult, dtype=torch.float32, device=st.device)
        self.seg_wd = torch.tensor(wd, dtype=torch.float32, device=st.device)

    def _hp_values(self):
        t = self.iterations + 1
        corr = math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)
        return [self.current_lr(), corr, self.grad_scale, self.clipvalue if self.clipvalue else 0.0]

    def apply_gradients(self):
        st = self.store
        if not self._prepared:
            self.prepare_step()
        self._prepared = False
        seg_sq, cn, tot, gn = self._clip_tables()
        _hip.call("iseg_adamw_step", K.ptr(st.flat_w), K.ptr(st.flat_g), K.ptr(self.m), K.ptr(self.v),
                  K.ptr(self.vhat) if self.amsgrad else None, K.ptr(st.flat_bf16), K.ptr(st.seg_of_block), K.ptr(self.seg_lr_mult),
                  K.ptr(self.seg_wd), K.ptr(self.hp), self.beta_1, self.beta_2, self.epsilon, K.ptr(seg_sq), cn, K.ptr(tot), gn, st.nblocks,
                  K.stream())
        self.iterations += 1
        nn.weights_changed()      # the step kernel rewrote the bf16 compute copies


class SGD(_FlatOptimizer):
    def __init__(self, learning_rate=0.01, momentum=0.0, nesterov=False, clipnorm=None, clipvalue=None, global_clipnorm=None, name="SGD"):
        super().__init__(learning_rate, clipnorm, clipvalue, global_clipnorm)
        self.nesterov = bool(nesterov)
        self.momentum = momentum
        self.l2_of = {}      # id(param) -> l2 coefficient, filled by utils.keras_ops.set_weight_decay

    def _build_tables(self):
        st = self.store
        self.m = getattr(self, "m", None) if getattr(self, "m", None) is not None else torch.zeros_like(st.flat_w)
        lr_mult = [float(getattr(p, "lr_multiplier", 1.0)) if (p.requires_grad and self._owns(p)) else 0.0 for p in st.params]
        l2 = [float(getattr(p, "l2_regularizer", 0.0)) if (p.requires_grad and self._owns(p)) else 0.0 for p in st.params]
        self.seg_lr_mult = torch.tensor(lr_mult, dtype=torch.float32, device=st.device)
        self.seg_l2 = torch.tensor(l2, dtype=torch.float32, device=st.device)

    def _hp_values(self):
        return [self.current_lr(), 0.0, self.grad_scale, self.clipvalue if self.clipvalue else 0.0]

    def apply_gradients(self):
        st = self.store
        if not self._prepared:
            self.prepare_step()
        self._prepared = False
        seg_sq, cn, tot, gn = self._clip_tables(self.seg_l2)
        _hip.call("iseg_sgd_momentum_step", K.ptr(st.flat_w), K.ptr(st.flat_g), K.ptr(self.m), K.ptr(st.flat_bf16),
                  K.ptr(st.seg_of_block), K.ptr(self.seg_lr_mult), K.ptr(self.seg_l2), K.ptr(self.hp), self.momentum, int(self.nesterov),
                  K.ptr(seg_sq), cn, K.ptr(tot), gn, st.nblocks, K.stream())
        self.iterations += 1
        nn.weights_changed()      # the step kernel rewrote the bf16 compute copies
