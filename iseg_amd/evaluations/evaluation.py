"""evaluations/evaluation.py of the reference (:19-144): multi-scale + flip inference over a validation set with the running
ignore-label loss and mean IoU -- evaluate(distribute_strategy, model, data, batch_size, num_class, ...) -> mean IoU."""
import torch

from .. import dist, nn
from ..core_model import SegFoundation
from ..losses.catecrossentropy_ignore_label import catecrossentropy_ignore_label_loss
from ..metrics.mean_iou import MeanIOU, per_class_miou_to_mean_miou
from ..metrics.seg_metric_wrapper import SegMetricWrapper


class _Mean:
    """keras.metrics.Mean over every element handed to update_state; replicas are summed when the result is read"""

    def __init__(self, name="loss"):
        self.name = name
        self.reset_state()

    def update_state(self, values):
        v = values.detach().float()
        self.total = self.total + v.sum().reshape(1)
        self.count += v.numel()

    def local_result(self):
        """this rank's running mean, no exchange (progress lines)"""
        return float(self.total) / max(float(self.count), 1.0)

    def result(self):
        """COLLECTIVE: every rank must call it the same number of times"""
        t = torch.cat([self.total.to(nn.device()), torch.tensor([float(self.count)], device=nn.device())])
        dist.all_reduce_sum(t)
        return float(t[0]) / max(float(t[1]), 1.0)

    def reset_state(self):
        self.total, self.count = torch.zeros(1, device=nn.device()), 0


def prepare_dataset(distribute_strategy, data, batch_size=16, val_image_count=0):
    """(:126-144) batch without dropping the remainder; under data parallelism every rank takes its interleaved share of the BATCHES
    (tf.data's AutoShardPolicy.DATA on a distributed dataset)"""
    ds = data.batch(batch_size, drop_remainder=False)
    if dist.world_size() > 1:
        ds = ds.shard(dist.world_size(), dist.rank())
    return ds.prefetch(2, device=nn.device())


@torch.no_grad()
def eval_step(ds_inputs, model, scale_rates, flip, loss_func, loss_metrics, metrics, distribute_strategy=None):
    """(:98-123) multi-scale + flip logits -> per-position loss into the running mean, logits into every metric"""
    images, labels = ds_inputs
    predictions = model.inference_with_multi_scales(images, training=False, scale_rates=scale_rates, flip=flip)
    loss_metrics.update_state(loss_func(labels, predictions))
    for metric in metrics:
        metric.update_state(labels, predictions)
    return predictions


def evaluate(distribute_strategy, model, data, batch_size, num_class, ignore_label=255, scale_rates=[0.5, 0.75, 1.0, 1.25, 1.5, 1.75],
             flip=True, val_image_count=0, pre_compute_fns=[], verbose=1):
    if not isinstance(model, SegFoundation):
        raise ValueError("ALl model must based on SegFoundation")
    ds = prepare_dataset(distribute_strategy, data, batch_size, val_image_count=val_image_count)
    processed_count = 0
    with distribute_strategy.scope():
        loss_func = catecrossentropy_ignore_label_loss(num_class=num_class, ignore_label=ignore_label, batch_size=batch_size, reduction=False)
        loss_metrics = _Mean("loss")
        iou_metrics = SegMetricWrapper(MeanIOU(num_class), num_class=num_class, ignore_label=ignore_label, name="IOU")
        if pre_compute_fns is not None and isinstance(pre_compute_fns, list):
            for fn in pre_compute_fns:
                iou_metrics.add_pre_compute_fn(fn)
        for inputs in ds:
            eval_step(inputs, model, scale_rates, flip, loss_func, loss_metrics, [iou_metrics], distribute_strategy)
            processed_count += int(inputs[0].shape[0])
            if verbose and dist.rank() == 0:
                # the progress line shows THIS rank's running values: ranks may hold different batch counts after ds.shard, so nothing
                # inside the loop may be a collective (the reference prints the replica-merged value; the final figures below are merged)
                print("Processed : {:}, current loss = {:4f}, current IOU = {:.2f} %".format(processed_count, loss_metrics.local_result(),
                                                                                            float(iou_metrics.metric.local_result()) * 100))
        # collectives: entered by every rank, once each, in this order
        mean_loss = loss_metrics.result()
        per_class = iou_metrics.metric.per_class_result()
        mean_iou = per_class_miou_to_mean_miou(*per_class)
        if dist.rank() == 0:
            print("-----------------------------------------------")
            print(f"Mean loss on val set : {mean_loss}")
            print(f"Mean IoU on val set : {float(mean_iou)}")
            print("-----------------------------------------------")
            print("Per-class IoU on val set :")
            print(per_class)
        evaluate.last_mean_loss = mean_loss
        loss_metrics.reset_state()
        iou_metrics.reset_states()
    return mean_iou
