"""metrics/utils.py of the reference (:12-62): SegMetricBuilder."""
from .mean_iou import MeanIOU
from .seg_metric_wrapper import SegMetricWrapper


class SegMetricBuilder:
    def __init__(self, num_class, ignore_label):
        self.num_class = num_class
        self.ignore_label = ignore_label
        self.__metrics = []

    def add(self, prefix="", use_iou=True, pre_compute_fn=None, custom_metric_fns_list=[]):
        metrics_list = []
        if prefix is None:
            prefix = ""
        if prefix != "":
            prefix = prefix + "_"
        if use_iou:
            iou_metric = SegMetricWrapper(MeanIOU(self.num_class), num_class=self.num_class, ignore_label=self.ignore_label,
                                          name=prefix + "IOU")
            iou_metric.add_pre_compute_fn(pre_compute_fn)
            metrics_list.append(iou_metric)
        if custom_metric_fns_list is not None:
            if not isinstance(custom_metric_fns_list, list):
                custom_metric_fns_list = [custom_metric_fns_list]
            for fn in custom_metric_fns_list:
                m = fn(num_class=self.num_class, ignore_label=self.ignore_label, name=prefix)
                if isinstance(m, SegMetricWrapper):
                    m.add_pre_compute_fn(pre_compute_fn)
                metrics_list.append(m)
        self.__metrics.append(metrics_list)

    @property
    def metrics(self):
        return self.__metrics

    def to_dict(self, name_fn):
        return {name_fn(i): ml for i, ml in enumerate(self.__metrics)}
