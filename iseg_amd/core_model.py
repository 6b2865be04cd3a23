"""core_model.py of the reference: SegModelInferenceConfig :24-47, SegBase :51-326 (inference entry points),
SegFoundation :329-605 (loss / loss-weight / metric plumbing, inputs_process)."""
import numpy as np
import torch

from . import functional as F
from .core_inference import inference_fn
from .utils.common import get_scaled_size, resize_image
from .losses.catecrossentropy_ignore_label import catecrossentropy_ignore_label_loss
from .metrics.utils import SegMetricBuilder
from .nn import Layer


class SegModelInferenceConfig(object):
    def __init__(self, scale_rates=[1.0], flip=False, use_cpu_cache=False, resize_method="bilinear"):
        self.scale_rates = scale_rates
        self.flip = flip
        self.use_cpu_cache = use_cpu_cache
        self.resize_method = resize_method

    def to_dict(self):
        return {"scale_rates": self.scale_rates, "flip": self.flip, "use_cpu_cache": self.use_cpu_cache,
                "resize_method": self.resize_method}


class SegBase(Layer):
    def __init__(self, num_class=21, input_norm_type=None, inference_configs: SegModelInferenceConfig = None, **kwargs):
        super().__init__(**kwargs)
        self.num_class = num_class
        self.inference_sliding_window_size = None
        self.input_norm_type = input_norm_type
        if inference_configs is None:
            inference_configs = SegModelInferenceConfig()
        self.inference_configs = inference_configs

    def inference(self, inputs, training=False):
        return inference_fn(inputs, model=self, num_class=self.num_class, training=training,
                            sliding_window_crop_size=self.inference_sliding_window_size)

    def predict_step(self, data):
        x = data[0] if isinstance(data, (tuple, list)) else data
        return self.inference_with_multi_scales(x, training=False, **self.inference_configs.to_dict())

    # -- multi-scale / flip test-time augmentation (:128-326) --------------------------------------------------
    def inference_with_scale_inputs_process(self, inputs, training=False, scale_rate=1.0, flip=False, resize_method="bilinear"):
        x = inputs[0] if isinstance(inputs, (list, tuple)) else inputs
        if flip:
            x = F.flip_left_right(x)
        sizes = get_scaled_size(x, scale_rate, pad_mode=1)
        if x.dtype != torch.float32:
            x = F.cast_to(x, torch.float32)
        return resize_image(x, sizes, method=resize_method)

    def inference_with_scale(self, inputs, training=False, scale_rate=1.0, flip=False, resize_method="bilinear"):
        first = inputs[0] if isinstance(inputs, (list, tuple)) else inputs
        original_size = [int(first.shape[1]), int(first.shape[2])]
        x = self.inference_with_scale_inputs_process(inputs, training=training, scale_rate=scale_rate, flip=flip,
                                                     resize_method=resize_method)
        sizes = [int(x.shape[1]), int(x.shape[2])]
        window = self.inference_sliding_window_size
        if window is not None:
            window = (min(int(window[0]), sizes[0]), min(int(window[1]), sizes[1]))
        logits = inference_fn(x, model=self, num_class=self.num_class, training=training, sliding_window_crop_size=window)
        logits = resize_image(logits, original_size, method=resize_method)
        if flip:
            logits = F.flip_left_right(logits)
        return logits

    def inference_with_multi_scales(self, inputs, training=False, scale_rates=[1.0], flip=False, use_cpu_cache=False,
                                    resize_method="bilinear"):
        """sum over scales (and over the mirrored image when flip) of the logits resized back to the input size, divided by the
        number of passes; use_cpu_cache is accepted for signature parity -- 288 GB of HBM never needs the host round trip"""
        from . import kernels as K

        divide_factor = len(scale_rates) * (2 if flip else 1)
        total = None
        for mirrored in ([False, True] if flip else [False]):
            x = F.flip_left_right(inputs) if mirrored else inputs
            part = None
            for rate in scale_rates:
                logits = self.inference_with_scale(x, training=training, scale_rate=float(rate), flip=False, resize_method=resize_method)
                part = logits if part is None else K.axpby(part, logits, 1.0, 1.0)
            if mirrored:
                part = F.flip_left_right(part)
            total = part if total is None else K.axpby(total, part, 1.0, 1.0)
        return K.axpby(total, None, 1.0 / divide_factor, 0.0)


class SegFoundation(SegBase):
    def __init__(self, num_class=21, input_norm_type=None, custom_main_loss_fn=None, custom_main_metric_fn=None, num_aux_loss=0,
                 aux_loss_rate=0.4, aux_metric_names=None, aux_metric_iou_masks=None, aux_metric_pre_fns=[], use_ohem=False,
                 ohem_thresh=0.7, label_as_inputs=False, custom_aux_loss_fns=[], custom_aux_metrics_fns=[], use_focal_loss=False,
                 focal_loss_gamma=2.0, focal_loss_alpha=1.0, class_weights=None, **kwargs):
        super().__init__(num_class=num_class, input_norm_type=input_norm_type, **kwargs)
        self.custom_main_loss_fn = custom_main_loss_fn
        self.custom_main_metric_fn = custom_main_metric_fn
        assert num_aux_loss >= 0, f"num_aux_loss must >= 0, found {num_aux_loss}"
        self.num_aux_loss = num_aux_loss
        if isinstance(aux_loss_rate, tuple):
            aux_loss_rate = list(aux_loss_rate)
        if not isinstance(aux_loss_rate, list):
            aux_loss_rate = [aux_loss_rate] * num_aux_loss
        assert len(aux_loss_rate) == num_aux_loss, "aux_loss_rate must be scalar or has length = num_aux_loss"
        if num_aux_loss == 0:
            aux_metric_names = None
        assert (aux_metric_names is None) or (len(aux_metric_names) == num_aux_loss)
        self.aux_loss_rate = aux_loss_rate
        self.use_ohem = use_ohem
        self.ohem_thresh = ohem_thresh
        self.aux_metric_names = aux_metric_names
        self.aux_metric_iou_masks = aux_metric_iou_masks
        self.aux_metric_pre_fns = aux_metric_pre_fns
        self.label_as_inputs = label_as_inputs
        self.custom_aux_loss_fns = custom_aux_loss_fns
        self.custom_aux_metrics_fns = custom_aux_metrics_fns
        self.use_focal_loss = use_focal_loss
        self.focal_loss_gamma = focal_loss_gamma
        self.focal_loss_alpha = focal_loss_alpha
        self.model_class_weights = class_weights
        if use_ohem:
            raise NotImplementedError("OHEM is marked 'WIP DO NOT USE' in the reference (losses/ohem.py:6) and is out of scope")

    def inputs_process(self, image, label):
        is_label_collection = isinstance(label, (list, tuple, dict))
        if self.label_as_inputs:
            if is_label_collection:
                if isinstance(label, list):
                    label = tuple(label)
                if isinstance(label, tuple):
                    image = (image, *label)
                if isinstance(label, dict):
                    _image = image
                    image = label.copy()
                    image["image"] = _image
            else:
                image = (image, label)
        if self.num_aux_loss > 0:
            expected_num_outputs = self.num_aux_loss + 1
            if is_label_collection:
                if isinstance(label, dict):
                    label = list(label.values())
                if isinstance(label, list):
                    label = tuple(label)
            else:
                label = tuple([label] * expected_num_outputs)
        return image, label

    def _index_to_output_key(self, index):
        return f"output_{index + 1}"

    def add_class_weights(self, class_weights=None, new_class_weights=None):
        if new_class_weights is not None:
            new_class_weights = np.array(new_class_weights)
            if class_weights is not None:
                class_weights *= new_class_weights
            else:
                class_weights = new_class_weights
        return class_weights

    def custom_losses(self, num_class, ignore_label, batch_size, class_weights=None, reduction=False, **kwargs):
        class_weights = self.add_class_weights(new_class_weights=class_weights)
        class_weights = self.add_class_weights(class_weights=class_weights, new_class_weights=self.model_class_weights)
        common_kwargs = {"num_class": num_class, "ignore_label": ignore_label, "batch_size": batch_size, "reduction": reduction,
                         "class_weights": class_weights}

        def default_ce_loss(post_func):      # core_model.py:498-505 of the reference
            return catecrossentropy_ignore_label_loss(post_compute_fn=post_func, use_focal_loss=self.use_focal_loss,
                                                      focal_loss_gamma=self.focal_loss_gamma, focal_loss_alpha=self.focal_loss_alpha,
                                                      **common_kwargs, **kwargs)

        if self.custom_main_loss_fn is not None:
            loss_dict = {self._index_to_output_key(0): self.custom_main_loss_fn(**common_kwargs, **kwargs)}
        else:
            loss_dict = {self._index_to_output_key(0): default_ce_loss(None)}
        if self.custom_aux_loss_fns is None or len(self.custom_aux_loss_fns) == 0:
            for i in range(self.num_aux_loss):
                loss_dict[self._index_to_output_key(i + 1)] = default_ce_loss(None)
        else:
            assert len(self.custom_aux_loss_fns) == self.num_aux_loss
            for i in range(self.num_aux_loss):
                if self.custom_aux_loss_fns[i] is not None:
                    loss = self.custom_aux_loss_fns[i](**common_kwargs, **kwargs)
                else:
                    loss = default_ce_loss(None)
                loss_dict[self._index_to_output_key(i + 1)] = loss
        return loss_dict

    def custom_losses_weights(self):
        weights_dict = {self._index_to_output_key(0): 1.0}
        for i in range(self.num_aux_loss):
            weights_dict[self._index_to_output_key(i + 1)] = self.aux_loss_rate[i]
        return weights_dict

    def custom_metrics(self, num_class, ignore_label):
        metrics = SegMetricBuilder(num_class, ignore_label)
        fns = self.custom_main_metric_fn
        if fns is None:
            fns = []
        if isinstance(fns, tuple):
            fns = list(fns)
        if not isinstance(fns, list):
            fns = [fns]
        metrics.add(custom_metric_fns_list=fns)
        masks = self.aux_metric_iou_masks
        if masks is None or len(masks) == 0:
            masks = [False] * self.num_aux_loss
        pre_fns = self.aux_metric_pre_fns
        if pre_fns is None or len(pre_fns) == 0:
            pre_fns = [None] * self.num_aux_loss
        aux_fns = self.custom_aux_metrics_fns
        if aux_fns is None or len(aux_fns) == 0:
            aux_fns = [[]] * self.num_aux_loss
        for i in range(self.num_aux_loss):
            prefix = "aux" if self.aux_metric_names is None else self.aux_metric_names[i]
            metrics.add(f"{prefix}_{i}", use_iou=masks[i], pre_compute_fn=pre_fns[i], custom_metric_fns_list=aux_fns[i])
        return metrics.to_dict(self._index_to_output_key)

    def multi_optimizers_layers(self):
        return None
