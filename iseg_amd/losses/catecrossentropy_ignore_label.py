"""losses/catecrossentropy_ignore_label.py of the reference (:14-90): factory returning weighted_loss(y_true, y_pred).

weighted_loss returns the per-position loss vector [N*H*W] (Reduction.NONE) exactly as the reference does; Keras' loss
wrapper then averages it over ALL positions.  The returned callable also carries `.fused_mean(y_true, y_pred, weight)`,
which CoreTrain uses to obtain that mean and d(mean)/d(logits) from one fused kernel pass."""
import torch

from .. import functional as F
from .. import nn


def catecrossentropy_ignore_label_loss(num_class=21, ignore_label=255, class_weights=None, batch_size=2, reduction=False,
                                       pre_compute_fn=None, post_compute_fn=None, from_logits=True, use_focal_loss=False,
                                       focal_loss_gamma=2.0, focal_loss_alpha=0.25):
    # use_focal_loss: keras CategoricalFocalCrossentropy(alpha, gamma, from_logits) replaces the plain CE (:27-37)
    focal = (float(focal_loss_alpha), float(focal_loss_gamma)) if use_focal_loss else None
    if not from_logits:
        raise NotImplementedError("from_logits=False is not used by the reference's training path")
    cw = None
    if class_weights is not None and len(class_weights) > 0:
        assert len(class_weights) == num_class
        cw = torch.as_tensor(list(class_weights), dtype=torch.float32, device=nn.device())

    def weighted_loss(y_true, y_pred):
        local_batch_size = y_pred.shape[0]
        if pre_compute_fn is not None:
            y_true, y_pred = pre_compute_fn(y_true, y_pred)
        loss_value = F.softmax_ce_per_pixel(y_pred, y_true, num_class, ignore_label, cw, focal)
        if post_compute_fn is not None:
            loss_value = post_compute_fn(None, y_pred, loss_value, local_batch_size)
        if reduction:
            loss_value = loss_value.sum() / batch_size      # tf.nn.compute_average_loss
        return loss_value

    def fused_mean(y_true, y_pred, weight=1.0, cm=None):
        if pre_compute_fn is not None:
            y_true, y_pred = pre_compute_fn(y_true, y_pred)
        return F.softmax_ce_mean(y_pred, y_true, num_class, ignore_label, cw, weight, focal, cm)

    def fused_upsample_mean(y_true, deferred, weight=1.0, cm=None):
        return F.upsample_softmax_ce_mean(deferred, y_true, num_class, ignore_label, cw, weight, cm)

    weighted_loss.fused_mean = fused_mean if (post_compute_fn is None and not reduction) else None
    # CoreTrain's step may hand the low-resolution logits over (F.DeferredLogits): upsample + loss + gradient + confusion in one kernel
    weighted_loss.fused_upsample_mean = fused_upsample_mean if (post_compute_fn is None and pre_compute_fn is None and not reduction and
                                                                focal is None) else None
    weighted_loss.num_class = num_class
    # the trainer may let the loss kernel also update a MeanIOU confusion matrix built for the same classes / ignore label
    weighted_loss.confusion_spec = (num_class, ignore_label) if (pre_compute_fn is None and focal is None) else None
    return weighted_loss
