"""layers/normalizations.py of the reference (:14-23, :34-36, :39-132): the `normalization()` factory whose global
default is SYNC_BATCH_NORM with momentum 0.9 / epsilon 1e-3."""
import functools

from .base_layers import BatchNormalization

SyncBatchNormalization = functools.partial(BatchNormalization, synchronized=True)

GLOBAL = "global"
BATCH_NORM = "batch_norm"
SYNC_BATCH_NORM = "sync_batch_norm"
GROUP_NROM = "group_norm"   # (sic) spelling kept from the reference


def global_norm_method():
    return SYNC_BATCH_NORM


def normalization(axis=-1, momentum=0.9, epsilon=1e-3, center=True, scale=True, beta_initializer="zeros",
                  gamma_initializer="ones", moving_mean_initializer="zeros", moving_variance_initializer="ones",
                  beta_regularizer=None, gamma_regularizer=None, beta_constraint=None, gamma_constraint=None, groups=16,
                  method=GLOBAL, trainable=True, name=None, **kwargs):
    if method == GLOBAL or method is None:
        method = global_norm_method()
    common = dict(axis=axis, momentum=momentum, epsilon=epsilon, center=center, scale=scale, beta_initializer=beta_initializer,
                  gamma_initializer=gamma_initializer, moving_mean_initializer=moving_mean_initializer,
                  moving_variance_initializer=moving_variance_initializer, trainable=trainable)
    if method == BATCH_NORM:
        return BatchNormalization(synchronized=False, name=name if name is not None else BATCH_NORM, **common)
    if method == SYNC_BATCH_NORM:
        return SyncBatchNormalization(name=name if name is not None else BATCH_NORM, **common)
    if method == GROUP_NROM:
        from .groupnorm import GroupNormalization

        return GroupNormalization(groups=groups, axis=axis, epsilon=epsilon, center=center, scale=scale, trainable=trainable,
                                  name=name if name is not None else GROUP_NROM)
    raise ValueError("Not support norm mathod = {}".format(method))
