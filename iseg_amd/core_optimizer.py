"""core_optimizer.py of the reference (:18-187): get_optimizer(...) -> one optimizer, or a list of them when any keyword is given as a
list / tuple (one optimizer per position; scalars are shared), with a poly (WarmUpPolyDecay) or cosine schedule and the
sgd / adam / amsgrad / adamw families of optimizers/modern."""
import inspect

from .optimizers import modern as modern_optimizers
from .optimizers.polydecay import CosineDecay, WarmUpPolyDecay


def _schedule(initial_lr, end_lr, steps, warmup_steps, warmup_lr, decay_strategy, poly_decay_power):
    if decay_strategy == "poly":
        return WarmUpPolyDecay(initial_lr, steps, end_learning_rate=end_lr, power=poly_decay_power, warmup_steps=warmup_steps,
                               warmup_learning_rate=warmup_lr)
    if decay_strategy == "cosine":      # keras CosineDecay: with a warm-up it starts at warmup_lr and targets initial_lr (:139-154)
        start, target = (warmup_lr, initial_lr) if warmup_steps > 0 else (initial_lr, None)
        return CosineDecay(start, steps, alpha=float(end_lr) / float(initial_lr), warmup_steps=warmup_steps, warmup_target=target)
    return initial_lr      # any other strategy keeps the constant rate, as the reference does


_FAMILIES = {
    "sgd": lambda lr, o: modern_optimizers.SGD(learning_rate=lr, momentum=o["sgd_momentum_rate"], clipnorm=o["clipnorm"], clipvalue=o["clipvalue"]),
    "adam": lambda lr, o: modern_optimizers.AdamW(weight_decay=0.0, learning_rate=lr, amsgrad=False, clipnorm=o["clipnorm"], clipvalue=o["clipvalue"]),
    "amsgrad": lambda lr, o: modern_optimizers.AdamW(weight_decay=0.0, learning_rate=lr, amsgrad=True, clipnorm=o["clipnorm"], clipvalue=o["clipvalue"]),
    "adamw": lambda lr, o: modern_optimizers.AdamW(weight_decay=o["adamw_weight_decay"], learning_rate=lr, clipnorm=o["clipnorm"],
                                                   clipvalue=o["clipvalue"]),
}


def _one_optimizer(o):
    lr = _schedule(o["initial_lr"], o["end_lr"], o["epoch_steps"] * o["train_epoch"], o["warmup_steps"], o["warmup_lr"], o["decay_strategy"],
                   o["poly_decay_power"])
    make = _FAMILIES.get(o["optimizer"])
    if make is None:
        raise ValueError(f"Unsupported optimizer {o['optimizer']}")
    with o["distribute_strategy"].scope():
        return make(lr, o)


def _fan_out(options):
    """scalars are shared, sequences give one value per optimizer; a one-element sequence counts as a scalar.  Returns None when
    nothing asks for more than one optimizer, else the list of per-optimizer option dicts."""
    seqs = {k: list(v) for k, v in options.items() if isinstance(v, (list, tuple))}
    for k, v in seqs.items():
        if not v:
            raise AssertionError(f"empty sequence for optimizer option {k!r}")
    for k, v in seqs.items():
        if len(v) == 1:
            options[k] = v[0]
    lengths = {len(v) for v in seqs.values() if len(v) > 1}
    if not lengths:
        return None
    if len(lengths) > 1:
        a, b = sorted(lengths)[:2]
        raise ValueError(f"kwargs for optimizer must be scaler or list/tuple with same length, found ({a} vs {b})")
    n = lengths.pop()
    return [{k: (seqs[k][i] if k in seqs and len(seqs[k]) > 1 else options[k]) for k in options} for i in range(n)]


def get_optimizer(distribute_strategy, initial_lr=0.007, end_lr=0.0, epoch_steps=1000, train_epoch=30, warmup_steps=0, warmup_lr=0.0,
                  decay_strategy="poly", poly_decay_power=0.9, optimizer="sgd", sgd_momentum_rate=0.9, adamw_weight_decay=0.0001,
                  clipnorm=None, clipvalue=None):
    names = list(inspect.signature(get_optimizer).parameters)
    given = locals()
    options = {k: given[k] for k in names}
    print("Optimizer info : **********************")
    print({k: v for k, v in options.items() if k != "distribute_strategy"})
    each = _fan_out(options)
    if each is None:
        return _one_optimizer(options)
    return [_one_optimizer(o) for o in each]
