"""core_optimizer.py of the reference (:18-187): get_optimizer -- scalar-or-list kwargs -> one or several optimizers, poly / cosine
schedule, sgd / adam / amsgrad / adamw."""
from .optimizers import modern as modern_optimizers
from .optimizers.polydecay import CosineDecay, WarmUpPolyDecay


def get_optimizer(distribute_strategy, initial_lr=0.007, end_lr=0.0, epoch_steps=1000, train_epoch=30, warmup_steps=0, warmup_lr=0.0,
                  decay_strategy="poly", poly_decay_power=0.9, optimizer="sgd", sgd_momentum_rate=0.9, adamw_weight_decay=0.0001,
                  clipnorm=None, clipvalue=None):
    kwargs = {"distribute_strategy": distribute_strategy, "initial_lr": initial_lr, "end_lr": end_lr, "epoch_steps": epoch_steps,
              "train_epoch": train_epoch, "warmup_steps": warmup_steps, "warmup_lr": warmup_lr, "decay_strategy": decay_strategy,
              "poly_decay_power": poly_decay_power, "optimizer": optimizer, "sgd_momentum_rate": sgd_momentum_rate,
              "adamw_weight_decay": adamw_weight_decay, "clipnorm": clipnorm, "clipvalue": clipvalue}
    print("Optimizer info : **********************")
    print({k: v for k, v in kwargs.items() if k != "distribute_strategy"})
    keys = kwargs.keys()
    max_list_size = 0
    for key in keys:
        value = kwargs[key]
        if isinstance(value, (list, tuple)):
            value = list(value)
            list_size = len(value)
            kwargs[key] = value
            assert list_size > 0
            if list_size == 1:
                kwargs[key] = value[0]
            elif list_size >= max_list_size:
                max_list_size = list_size
            else:
                raise ValueError(f"kwargs for optimizer must be scaler or list/tuple with same length, found ({list_size} vs {max_list_size})")
    if max_list_size <= 1:
        return _get_optimizer(**kwargs)
    optimizer_list = []
    for i in range(max_list_size):
        sub = {k: (v[i] if isinstance(v, list) else v) for k, v in kwargs.items()}
        optimizer_list += [_get_optimizer(**sub)]
    return optimizer_list


def _get_optimizer(distribute_strategy, initial_lr=0.007, end_lr=0.0, epoch_steps=1000, train_epoch=30, warmup_steps=0,
                   warmup_lr=0.003, decay_strategy="poly", poly_decay_power=0.9, optimizer="sgd", sgd_momentum_rate=0.9,
                   adamw_weight_decay=0.0001, clipnorm=None, clipvalue=None):
    learning_rate = initial_lr
    steps = epoch_steps * train_epoch
    if decay_strategy == "poly":
        learning_rate = WarmUpPolyDecay(learning_rate, steps, end_learning_rate=end_lr, power=poly_decay_power,
                                        warmup_steps=warmup_steps, warmup_learning_rate=warmup_lr)
    elif decay_strategy == "cosine":
        _initial_lr, warmup_target = learning_rate, None
        alpha = float(end_lr) / float(learning_rate)
        if warmup_steps > 0:
            _initial_lr, warmup_target = warmup_lr, learning_rate
        learning_rate = CosineDecay(_initial_lr, steps, alpha=alpha, warmup_steps=warmup_steps, warmup_target=warmup_target)
    with distribute_strategy.scope():
        if optimizer == "sgd":
            return modern_optimizers.SGD(learning_rate=learning_rate, momentum=sgd_momentum_rate, clipnorm=clipnorm, clipvalue=clipvalue)
        if optimizer == "adam":
            return modern_optimizers.AdamW(weight_decay=0.0, learning_rate=learning_rate, amsgrad=False, clipnorm=clipnorm, clipvalue=clipvalue)
        if optimizer == "amsgrad":
            return modern_optimizers.AdamW(weight_decay=0.0, learning_rate=learning_rate, amsgrad=True, clipnorm=clipnorm, clipvalue=clipvalue)
        if optimizer == "adamw":
            return modern_optimizers.AdamW(weight_decay=adamw_weight_decay, learning_rate=learning_rate, clipnorm=clipnorm, clipvalue=clipvalue)
        raise ValueError(f"Unsupported optimizer {optimizer}")
