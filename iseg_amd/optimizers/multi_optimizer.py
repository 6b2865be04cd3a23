"""optimizers/multi_optimizer.py of the reference (MultiOptimizer :10-107, after TensorFlow-Addons' discriminative layer training): several
optimizers, each responsible for the variables of its own layers; (gradient, variable) pairs are routed by variable NAME (:42-55).

Here every sub-optimizer is one of the flat-buffer optimizers (optimizers/modern.py) built over the SAME ParamStore: a sub-optimizer's
per-variable tables carry learning-rate multiplier 0 and no decay for the variables it does not own, so its single step kernel leaves them
untouched, and the sub-optimizers' kernels run one after the other (the reference chains their update ops the same way, :57-63).
`global_clipnorm` on a sub-optimizer would need the norm over its own variables only and raises; `clipnorm` / `clipvalue` act per variable
and are fine."""


class MultiOptimizer:
    def __init__(self, optimizers_and_layers=None, optimizer_specs=None, name="MultiOptimizer", **kwargs):
        self.name = name
        if optimizer_specs is None and optimizers_and_layers is not None:
            self.optimizer_specs = [self.create_optimizer_spec(opt, layers) for opt, layers in optimizers_and_layers]
        elif optimizer_specs is not None and optimizers_and_layers is None:
            self.optimizer_specs = [self.maybe_initialize_optimizer_spec(spec) for spec in optimizer_specs]
        else:
            raise RuntimeError("Must specify one of `optimizers_and_layers` or `optimizer_specs`.")
        self.grad_scale = 1.0
        self.store = None

    @classmethod
    def create_optimizer_spec(cls, optimizer, layers_or_model):
        """{"optimizer", "weights"}: the NAMES of the variables (parameters and state) of the layers, as in the reference (:71-91)"""
        layers = layers_or_model if isinstance(layers_or_model, list) else [layers_or_model]
        weights = []
        for layer in layers:
            for v in list(layer.parameters()) + list(layer.buffers()):
                n = getattr(v, "iseg_name", None)
                if n is not None and n not in weights:
                    weights.append(n)
        return {"optimizer": optimizer, "weights": weights}

    @classmethod
    def maybe_initialize_optimizer_spec(cls, optimizer_spec):
        if isinstance(optimizer_spec["optimizer"], dict):
            raise NotImplementedError("MultiOptimizer: serialised optimizer configs are not supported, pass optimizer objects")
        return optimizer_spec

    # ---- the flat-optimizer protocol of trainer.TrainableModel -------------------------------------------------
    def build(self, store):
        self.store = store
        owned = {}
        for k, spec in enumerate(self.optimizer_specs):
            opt = spec["optimizer"]
            if getattr(opt, "global_clipnorm", None):
                raise NotImplementedError("MultiOptimizer: global_clipnorm on a sub-optimizer (the norm would have to cover its own variables only)")
            for n in spec["weights"]:
                if n in owned:
                    raise ValueError(f"MultiOptimizer: variable {n} belongs to optimizers {owned[n]} and {k}")
                owned[n] = k
            opt.restrict_to(spec["weights"])
            opt.build(store)

    def apply_gradients(self, grads_and_vars=None, name=None, **kwargs):
        for spec in self.optimizer_specs:
            opt = spec["optimizer"]
            opt.grad_scale = self.grad_scale
            opt.apply_gradients()

    def exclude_from_weight_decay(self, var_list=None, var_names=None):
        for spec in self.optimizer_specs:
            fn = getattr(spec["optimizer"], "exclude_from_weight_decay", None)
            if callable(fn):
                fn(var_list=var_list, var_names=var_names)

    @property
    def iterations(self):
        return self.optimizer_specs[0]["optimizer"].iterations

    @iterations.setter
    def iterations(self, value):
        for spec in self.optimizer_specs:
            spec["optimizer"].iterations = value

    def current_lr(self):
        return self.optimizer_specs[0]["optimizer"].current_lr()

    def get_config(self):
        return {"name": self.name, "optimizer_specs": [{"optimizer": type(s["optimizer"]).__name__, "weights": list(s["weights"])}
                                                       for s in self.optimizer_specs]}

    def __repr__(self):
        return "Multi Optimizer with %i optimizer layer pairs" % len(self.optimizer_specs)
