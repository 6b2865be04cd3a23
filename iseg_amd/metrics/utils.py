"""SegMetricBuilder (reference metrics/utils.py:12-62): one list of metric objects per model output; `metrics` hands the lists to
compile() in output order and `to_dict` keys them by output name."""
from .mean_iou import MeanIOU
from .seg_metric_wrapper import SegMetricWrapper


def _as_list(fns):
    if fns is None:
        return []
    return list(fns) if isinstance(fns, (list, tuple)) else [fns]


class SegMetricBuilder:
    def __init__(self, num_class, ignore_label):
        self.num_class, self.ignore_label = num_class, ignore_label
        self._per_output = []

    def _make(self, factory, tag, pre_compute_fn):
        metric = factory(num_class=self.num_class, ignore_label=self.ignore_label, name=tag)
        if isinstance(metric, SegMetricWrapper):
            metric.add_pre_compute_fn(pre_compute_fn)
        return metric

    def add(self, prefix="", use_iou=True, pre_compute_fn=None, custom_metric_fns_list=None):
        """metrics of the next output: the running mean IoU (named '<prefix>_IOU') unless use_iou is off, then whatever the factories in
        custom_metric_fns_list build from (num_class, ignore_label, name='<prefix>_')"""
        tag = f"{prefix}_" if prefix else ""
        factories = _as_list(custom_metric_fns_list)
        if use_iou:
            factories.insert(0, lambda num_class, ignore_label, name: SegMetricWrapper(MeanIOU(num_class), num_class=num_class,
                                                                                      ignore_label=ignore_label, name=name + "IOU"))
        self._per_output.append([self._make(f, tag, pre_compute_fn) for f in factories])

    @property
    def metrics(self):
        return self._per_output

    def to_dict(self, name_fn):
        return {name_fn(i): group for i, group in enumerate(self._per_output)}
