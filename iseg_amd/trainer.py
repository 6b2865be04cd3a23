"""The train step Keras' Model.fit runs for the reference (core_train.py:141-152; SURVEY 3.3 "HOT LOOP"), written out:

    zero grads -> forward (training=True) -> sum_k w_k * mean(loss_k) -> backward (kernels write the flat gradient buffer,
    SyncBN all-reduces inside) -> bucketed gradient all-reduce over RCCL, overlapped with backward -> fused optimizer step
    (also refreshes the bf16 weight shadows) -> running confusion-matrix metrics.

Everything between the H2D copy of the batch and the optimizer step is enqueued on one HIP stream without a host sync;
loss values stay on the device until somebody asks for them.
"""
import torch

from . import dist
from . import functional as F
from . import kernels as K
from . import nn
from .param_store import ParamStore


class TrainableModel:
    def __init__(self, model, optimizer=None, loss=None, loss_weights=None, metrics=None, jit_compile=None):
        self.model = model
        self.optimizer = optimizer
        self.loss = loss
        self.loss_weights = loss_weights or {}
        self.metrics = metrics or {}
        self.update_metrics = True
        # jit_compile (Keras: compile the train function with XLA; core_train.py:86-91 passes it through model.compile): the MI355X counterpart of a
        # traced, compiled step is the step replayed from ONE HIP graph (iseg_amd/graphs.py).  True -> fit() replays; None / False -> eager launches.
        # The replay produces the bits of the eager step, so the switch changes speed only.
        self.jit_compile = bool(jit_compile)
        self._graphed_step = None
        # keras.optimizers.Optimizer(gradient_transformers=[...]): functions applied to the gradients before the update.  Here a transformer
        # takes the ParamStore (flat gradient buffer + per-parameter views) after the data-parallel sum and edits the gradients in place.
        self.gradient_transformers = []
        if isinstance(optimizer, list):
            # several optimizers: the model names the layer groups (core_model.py:603, layers/core_model_ext.py:386-388 multi_optimizers_layers),
            # one group per optimizer, and the pairs become a MultiOptimizer (optimizers/multi_optimizer.py)
            from .optimizers.multi_optimizer import MultiOptimizer

            fn = getattr(model, "multi_optimizers_layers", None)
            groups = fn() if callable(fn) else None
            if not groups or len(groups) != len(optimizer):
                raise ValueError(f"a list of {len(optimizer)} optimizers needs model.multi_optimizers_layers() to return as many layer groups "
                                 f"(got {None if not groups else len(groups)})")
            optimizer = self.optimizer = MultiOptimizer(optimizers_and_layers=[(o, g if isinstance(g, list) else [g])
                                                                                for o, g in zip(optimizer, groups)])
        if hasattr(model, "build_with_dummy"):
            model.build_with_dummy()
        params = list(model.parameters())
        self.store = getattr(model, "_iseg_store", None)
        if self.store is None or [id(p) for p in self.store.params] != [id(p) for p in _unique(params)]:
            self.store = ParamStore(params)
            model._iseg_store = self.store
        if dist.active():
            self.store.broadcast_from_rank0()
            for b in model.buffers():
                dist.broadcast(b, 0)
        self.reducer = dist.GradReducer(self.store)
        if optimizer is not None:
            optimizer.build(self.store)
        self.last_losses = None

    # ---- helpers ------------------------------------------------------------------------------------------
    def _key(self, i):
        return f"output_{i + 1}"

    def _loss_fn(self, i):
        if isinstance(self.loss, dict):
            return self.loss[self._key(i)]
        if isinstance(self.loss, (list, tuple)):
            return self.loss[i]
        return self.loss

    def _weight(self, i):
        if isinstance(self.loss_weights, dict):
            return float(self.loss_weights.get(self._key(i), 1.0))
        if isinstance(self.loss_weights, (list, tuple)):
            return float(self.loss_weights[i])
        return 1.0

    def _metrics_for(self, i):
        if isinstance(self.metrics, dict):
            return self.metrics.get(self._key(i), [])
        return self.metrics if i == 0 else []

    def __call__(self, x, training=False):
        return self.model(x, training=training)

    def _fusable_confusion(self, fn, i, y_true, out):
        """the MeanIOU confusion matrix of output i when the loss kernel can update it itself: exactly one SegMetricWrapper(MeanIOU)
        without pre-compute hooks, same class count / ignore label as the loss, labels already at the logits size (no nearest resize)"""
        from .metrics.mean_iou import MeanIOU
        from .metrics.seg_metric_wrapper import SegMetricWrapper

        spec = getattr(fn, "confusion_spec", None)
        ms = self._metrics_for(i)
        if not self.update_metrics or spec is None or len(ms) != 1:
            return None
        m = ms[0]
        if not isinstance(m, SegMetricWrapper) or m._pre_compute_fn_list or not isinstance(m.metric, MeanIOU):
            return None
        if (m.num_class, m.ignore_label) != spec or out.dim() != 4 or out.shape[-1] != m.num_class:
            return None
        if tuple(y_true.shape[1:3]) != tuple(out.shape[1:3]) or y_true.shape[0] != out.shape[0]:
            return None
        return m.metric.total_cm

    # ---- the hot loop ---------------------------------------------------------------------------------------
    def train_step(self, x, y):
        ys = y if isinstance(y, (tuple, list)) else None
        self.store.zero_grad()
        dist.set_active_reducer(self.reducer)
        with F.defer_logits_upsample(), F.drop_path_pool():
            outputs = self.model(x, training=True)
        if isinstance(outputs, dict):
            outputs = list(outputs.values())
        outputs = list(outputs) if isinstance(outputs, (list, tuple)) else [outputs]
        losses = []
        fused_metric_outputs = set()
        for i, out in enumerate(outputs):
            fn = self._loss_fn(i)
            yt = ys[i] if ys is not None else y
            w = self._weight(i)
            if isinstance(out, F.DeferredLogits):
                # low-resolution logits: one kernel upsamples, takes the loss and its gradient and updates the running mIoU -- when the
                # loss is the stock ignore-label CE and every metric of this output can ride along; otherwise materialise as usual
                up = getattr(fn, "fused_upsample_mean", None)
                cm = self._fusable_confusion(fn, i, yt, out)
                metrics_ride = (not self.update_metrics) or not self._metrics_for(i) or cm is not None
                if up is not None and metrics_ride and out.fusable(fn.num_class) and tuple(yt.shape[:3]) == tuple(out.shape[:3]):
                    fused_metric_outputs.add(i)
                    losses.append(up(yt, out, w, cm=cm))
                    continue
                outputs[i] = out = out.materialize()
            fused = getattr(fn, "fused_mean", None)
            if fused is not None:
                cm = self._fusable_confusion(fn, i, yt, out)
                if cm is not None:
                    fused_metric_outputs.add(i)
                    losses.append(fused(yt, out, w, cm=cm))      # loss, its gradient and the running-mIoU update in one pass
                else:
                    losses.append(fused(yt, out, w))
            else:
                lv = fn(yt, out)
                losses.append(lv.float().mean() * w)
        with F.unit_loss_grad(), K.deferred_reductions(self.store.flat_g):      # (flushed on exit, and before a gradient bucket goes out under data parallelism)
            torch.autograd.backward(losses)
        self.reducer.finish()
        dist.set_active_reducer(None)
        for fn in self.gradient_transformers:
            fn(self.store)
        self.optimizer.grad_scale = 1.0 / dist.world_size()    # per-replica mean losses, summed grads -> global mean
        self.optimizer.apply_gradients()
        if self.update_metrics:
            with torch.no_grad():
                for i, out in enumerate(outputs):
                    if i in fused_metric_outputs:
                        continue
                    yt = ys[i] if ys is not None else y
                    for m in self._metrics_for(i):
                        m.update_state(yt, out.detach())
        self.last_losses = losses
        return losses

    @torch.no_grad()
    def test_step(self, x, y):
        out = self.model.inference(x, training=False) if hasattr(self.model, "inference") else self.model(x, training=False)
        outs = out if isinstance(out, (list, tuple)) else [out]
        ys = y if isinstance(y, (tuple, list)) else None
        for i, o in enumerate(outs):
            yt = ys[i] if ys is not None else y
            for m in self._metrics_for(i):
                m.update_state(yt, o)
        return outs

    def reset_metrics(self):
        for i in range(8):
            for m in self._metrics_for(i):
                m.reset_states()

    def metric_results(self):
        res = {}
        for i in range(8):
            for m in self._metrics_for(i):
                res[f"{self._key(i)}_{m.name}"] = float(m.result())
        return res

    def fit(self, train_ds, epochs=1, validation_data=None, callbacks=(), initial_epoch=0, steps_per_epoch=1000, validation_steps=None,
            verbose=1, validation_freq=1, log_every=50):
        it = iter(train_ds)
        history = []
        step_fn = self.train_step
        if self.jit_compile:
            if self._graphed_step is None:
                from .graphs import GraphedTrainStep

                self._graphed_step = GraphedTrainStep(self)      # (falls back to the eager step where a capture is not possible: CPU, c10d data parallelism)
            step_fn = self._graphed_step
        for epoch in range(initial_epoch, epochs):
            for cb in callbacks:
                cb.on_epoch_begin(epoch)
            self.reset_metrics()
            running = None
            for step in range(steps_per_epoch):
                x, y = next(it)
                losses = step_fn(x, y)
                if verbose and (step + 1) % log_every == 0:
                    vals = [float(l) for l in losses]          # the only host sync, once per log interval
                    print(f"epoch {epoch} step {step + 1}/{steps_per_epoch} loss {sum(vals):.5f} lr {self.optimizer.current_lr():.3e}")
                running = losses
            logs = {"loss": float(sum(float(l) for l in running)) if running is not None else float("nan")}
            logs.update(self.metric_results())
            if validation_data is not None and (epoch + 1) % validation_freq == 0:
                self.reset_metrics()
                vit = iter(validation_data)
                n = validation_steps if validation_steps is not None else 10 ** 9
                for _ in range(n):
                    try:
                        vx, vy = next(vit)
                    except StopIteration:
                        break
                    self.test_step(vx, vy)
                logs.update({"val_" + k: v for k, v in self.metric_results().items()})
            history.append(logs)
            if verbose:
                print(f"epoch {epoch}: {logs}")
            for cb in callbacks:
                cb.on_epoch_end(epoch, logs)
        return history


def _unique(params):
    seen, out = set(), []
    for p in params:
        if id(p) not in seen:
            seen.add(id(p))
            out.append(p)
    return out
