"""metrics/seg_metric_wrapper.py of the reference (:22-110): nearest-resize the label map to the logits size, argmax
(first maximal index), drop ignored labels, accumulate."""
import torch

from .. import kernels as K


class SegMetricWrapper:
    def __init__(self, metric, num_class=21, ignore_label=255, name=None):
        self.name = name
        self.metric = metric
        self.num_class = num_class
        self.ignore_label = ignore_label
        self._pre_compute_fn_list = []

    def add_pre_compute_fn(self, fn):
        if fn is None:
            return
        self._pre_compute_fn_list.append(fn)

    def update_state(self, y_true, y_pred, sample_weight=None):
        for fn in self._pre_compute_fn_list:
            y_true, y_pred = fn(y_true, y_pred)
        y_true = y_true.to(torch.int32)
        if y_pred.dim() == 4:
            if y_true.dim() == 3:
                y_true = y_true.unsqueeze(-1)
            if tuple(y_true.shape[1:3]) != tuple(y_pred.shape[1:3]):
                y_true = K.resize_nearest_i32(y_true.contiguous(), y_pred.shape[1], y_pred.shape[2])
        z = y_pred.reshape(-1, self.num_class)
        if z.dtype != torch.float32:
            z = K.cast(z.contiguous(), torch.float32)
        y = y_true.reshape(-1).contiguous()
        ignore = self.ignore_label
        if ignore == 0:
            # reference: mask = y != 0, then y -= 1 (process_seg_metric_inputs :56-59)
            y = y - 1
            ignore = -1
        self.metric.update_from_logits(z.contiguous(), y, ignore)

    def result(self):
        return self.metric.result()

    def reset_states(self):
        self.metric.reset_states()
