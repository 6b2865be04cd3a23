"""HIP-graph replay of launch-bound inference calls.  No counterpart in the reference (its tf.function traces play this role):
a batch-1 sliding-window inference of ViT-B is ~250 launches of 5-20 us kernels, i.e. bound by the host's enqueue rate, not by the GPU.

GraphedCall(fn) runs fn(x) eagerly for the first calls of a given input signature (shape, dtype), then captures ONE replay graph on a side
stream and afterwards only copies the input into the captured buffer and replays.  Requirements on fn (all met by model(x, training=False)
and core_inference.inference_with_sliding_window): no host synchronisation, no host-side randomness, every kernel on torch's current stream
(iseg_amd.kernels.stream()), weights not re-homed between calls.

GraphedTrainStep(trainable) does the same for TrainableModel.train_step (round 3): ~400 launches per flagship step cost the Python host
8.3 ms to enqueue, against 9.5 ms of kernels -- the step is one kernel speed-up away from being host-bound, and the small-image configurations
(ResNet-50 at 256^2, InternImage) already are.  What the host used to decide per step lives in device memory instead: the optimizer's
scalars (one fixed slot, filled by a stream-ordered copy in front of the replay) and the dropout / drop-path draw counter (a one-word
addend of every frozen seed argument).  Data-parallel runs are captured when the exchange is stream-ordered (ISEG_DIST_NATIVE=1, iseg_amd/dist.py);
with c10d work objects they keep the eager step."""
import torch

from . import kernels as K


class GraphedCall:
    def __init__(self, fn, warmup=2):
        self.fn = fn
        self.warmup = int(warmup)
        self.entries = {}      # signature -> [calls so far, graph, static input, static output, buffer generation at capture]

    @staticmethod
    def _signature(x):
        return (tuple(x.shape), x.dtype, x.device)

    def _capture(self, x):
        side = torch.cuda.Stream(device=x.device)
        static_in = torch.empty_like(x)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            static_in.copy_(x)
            self.fn(static_in)      # on the capture stream: its workspace (keyed by stream) and every lazy table reach their final size here
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph, stream=side):
            out = self.fn(static_in)
        torch.cuda.current_stream().wait_stream(side)
        return graph, static_in, out

    @staticmethod
    def _refresh_derived_weights():
        """The K-contiguous kernel copies (nn.wt) and the ConvNeXt blocks' tiled / layer-scale-folded images (nn.mlp_tiled, nn.w_colscaled) are
        refreshed by HOST-side version checks; after the warm-up calls the versions match, so the capture holds none of those launches and a
        replay never passes through the checks again.  After an optimizer step / load_weights / restore_checkpoint they are refreshed here,
        eagerly, in front of the replay (same stream: ordered).  Returns the generation of the buffers behind them."""
        from . import nn

        nn.refresh_wt()
        nn.refresh_prep()
        return nn.buffers_generation()

    def __call__(self, x):
        if not (torch.is_tensor(x) and x.is_cuda):
            raise TypeError("GraphedCall takes one device tensor")
        sig = self._signature(x)
        e = self.entries.get(sig)
        if e is None:
            e = self.entries[sig] = [0, None, None, None, -1]
        if e[1] is not None and self._refresh_derived_weights() != e[4]:
            # the buffers the graph points into were re-allocated since the capture (a model dropped or built, a parameter registered):
            # drop the graph, run this call eagerly and capture again on the next one
            e[0], e[1], e[2], e[3] = self.warmup, None, None, None
            with torch.no_grad():
                return self.fn(x)
        if e[1] is None:
            e[0] += 1
            if e[0] <= self.warmup:      # eager: builds layers, transposed kernel copies, tiling plans
                with torch.no_grad():
                    return self.fn(x)
            e[1], e[2], e[3] = self._capture(x)
            e[4] = self._refresh_derived_weights()
        graph, static_in, out = e[1], e[2], e[3]
        static_in.copy_(x)
        graph.replay()
        return out      # (the captured output buffer: valid until the next call with this signature)


def graphed_inference(model, sliding_window_crop_size=None):
    """inference_fn (core_inference.py:46-57 of the reference) as a replayed graph: logits = graphed_inference(model, (512, 512))(images)"""
    from .core_inference import inference_fn

    return GraphedCall(lambda x: inference_fn(x, model, training=False, sliding_window_crop_size=sliding_window_crop_size))


class GraphedTrainStep:
    """train_step(x, y) of a trainer.TrainableModel replayed from ONE HIP graph per input signature.

    step = GraphedTrainStep(trainable);  losses = step(x, y)      # same contract as trainable.train_step

    The first `warmup` calls run eagerly (layers build, kernel copies and tiling plans reach their final size); the next call runs its step
    eagerly on the capture stream and then captures; from then on every call copies the batch into the captured buffers, pushes the
    optimizer's scalars, advances the draw counter and replays.  Replayed steps produce bit for bit what eager steps produce from the same
    state (same seeds, same summation orders): tests/test_graph_train_gpu.py.  Falls back to the eager step under data parallelism with c10d work objects (captured with the stream-ordered exchange, ISEG_DIST_NATIVE=1), for a
    MultiOptimizer, and on the CPU."""

    SEED_STRIDE = 0xD1B54A32D192ED03      # functional.next_seed(): seed = base + counter * SEED_STRIDE (+ rank term)

    def __init__(self, trainable, warmup=3):
        self.tm = trainable
        self.warmup = int(warmup)
        self.entries = {}

    def _eligible(self, x):
        from . import dist

        opt = self.tm.optimizer
        # data parallel: only with the stream-ordered exchange (dist.native_mode() == "rccl": every collective is one enqueue on an explicit
        # stream, so the step -- SyncBN messages, bucket all-reduces on the side stream -- is one graph); c10d work objects are host-driven
        capturable = not dist.active() or dist.native_mode() == "rccl"
        return torch.is_tensor(x) and x.is_cuda and capturable and hasattr(opt, "fixed_hp_slot") and opt.store is not None

    @staticmethod
    def _signature(x, y):
        ys = y if isinstance(y, (tuple, list)) else (y,)
        return (tuple(x.shape), x.dtype) + tuple((tuple(t.shape), t.dtype) for t in ys)

    def _capture(self, x, y):
        from . import functional as F

        tm, opt = self.tm, self.tm.optimizer
        side = torch.cuda.Stream(device=x.device)
        sx = torch.empty_like(x)
        sy = [torch.empty_like(t) for t in y] if isinstance(y, (tuple, list)) else torch.empty_like(y)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            sx.copy_(x)
            for d, s_ in zip(sy if isinstance(sy, list) else [sy], y if isinstance(y, (tuple, list)) else [y]):
                d.copy_(s_)
            eager_losses = tm.train_step(sx, sy)      # THIS call's step, run on the capture stream: its workspace (keyed by stream) and lazy tables reach their final size
        side.synchronize()
        seed_off = torch.zeros(1, dtype=torch.int64, device=x.device)
        opt.fixed_hp_slot(True)
        counter0 = F._RNG_COUNTER[0]
        it0 = opt.iterations
        graph = torch.cuda.CUDAGraph()
        K.set_seed_offset(seed_off)
        try:
            with torch.cuda.stream(side):
                opt.grad_scale = 1.0
                opt.prepare_step()      # outside the capture: the captured step kernel only reads the fixed slot
            with torch.cuda.graph(graph, stream=side):
                losses = tm.train_step(sx, sy)
        finally:
            K.set_seed_offset(None)
        draws = F._RNG_COUNTER[0] - counter0
        # nothing ran while capturing: take the host-side bookkeeping of that "step" back
        F._RNG_COUNTER[0] = counter0
        opt.iterations = it0
        opt._prepared = False
        torch.cuda.current_stream().wait_stream(side)
        from . import nn

        return dict(graph=graph, sx=sx, sy=sy, losses=losses, seed_off=seed_off, counter0=counter0, draws=draws, generation=nn.buffers_generation(),
                    off_host=torch.zeros(8, dtype=torch.int64).pin_memory(), off_events=[None] * 8, off_slot=0), eager_losses

    def __call__(self, x, y):
        if not self._eligible(x):
            return self.tm.train_step(x, y)
        from . import functional as F

        sig = self._signature(x, y)
        e = self.entries.get(sig)
        if e is None:
            e = self.entries[sig] = dict(calls=0, graph=None)
        if e["graph"] is None:
            e["calls"] += 1
            if e["calls"] <= self.warmup:
                return self.tm.train_step(x, y)
            captured, eager_losses = self._capture(x, y)      # this call's step ran eagerly inside; replays start with the next call
            e.update(captured)
            return eager_losses
        from . import nn

        if e["generation"] != nn.buffers_generation():
            # the derived-weight buffers / pointer tables the captured launches point into were re-allocated (another model built or dropped):
            # this step runs eagerly -- which also rebuilds them -- and the next call captures again
            self.entries[sig] = dict(calls=self.warmup, graph=None)
            return self.tm.train_step(x, y)
        return self._replay(e, x, y)

    def input_buffers(self, x, y):
        """(sx, sy): the captured step's own input tensors for this signature, or None before the capture.  A producer (the on-device input
        pipeline, data_process/pipeline.py) that writes the next batch straight into them and passes THEM to the call saves the per-step copy."""
        e = self.entries.get(self._signature(x, y))
        if e is None or e.get("graph") is None:
            return None
        return e["sx"], e["sy"]

    def _replay(self, e, x, y):
        from . import functional as F

        tm, opt = self.tm, self.tm.optimizer
        # a batch that already sits in the captured input buffers (input_buffers(): the input pipeline wrote it there) needs no copy -- the eager
        # step reads the caller's tensors in place too; the 50 MB image copy + the label copy were 36 us of every replayed flagship step
        if x.data_ptr() != e["sx"].data_ptr():
            e["sx"].copy_(x)
        if isinstance(e["sy"], list):
            for d, s_ in zip(e["sy"], y):
                if s_.data_ptr() != d.data_ptr():
                    d.copy_(s_)
        elif y.data_ptr() != e["sy"].data_ptr():
            e["sy"].copy_(y)
        # the draws of this step: the frozen seeds belong to counters counter0 + 1 .. counter0 + draws
        off = ((F._RNG_COUNTER[0] - e["counter0"]) * self.SEED_STRIDE) & 0xFFFFFFFFFFFFFFFF
        slot = e["off_slot"]
        e["off_slot"] = (slot + 1) % 8
        ev = e["off_events"][slot]
        if ev is not None:
            ev.synchronize()
        e["off_host"][slot] = off - (1 << 64) if off >= (1 << 63) else off
        e["seed_off"].copy_(e["off_host"][slot:slot + 1], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        e["off_events"][slot] = ev
        F._RNG_COUNTER[0] += e["draws"]
        opt.grad_scale = 1.0
        opt.prepare_step()
        e["graph"].replay()
        opt.after_replayed_step()
        tm.last_losses = e["losses"]
        return e["losses"]
