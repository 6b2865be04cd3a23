"""layers/core_model_ext.py of the reference (:25-402): SegManaged = backbone -> head -> 1x1 logits conv(s) -> bilinear
upsample to the input size -> float32.  The user sets `.head` (a callable taking the endpoints list)."""
import torch

from .. import functional as F
from .. import static_strings as ss
from ..backbones.feature_extractor import get_backbone
from ..core_model import SegFoundation
from ..utils.common import resize_image
from .base_layers import Conv2D


class SegManaged(SegFoundation):
    def __init__(self, backbone_name=ss.RESNET50, backbone_weights_path=None, backbone_custom_fn=None, output_stride=32, num_class=21,
                 input_norm_type=None, build_input_size=(512, 512), custom_main_loss_fn=None, num_aux_loss=0, aux_loss_rate=0.4,
                 aux_metric_names=None, custom_aux_loss_fns=[], use_ohem=False, ohem_thresh=0.7, use_focal_loss=False,
                 focal_loss_gamma=2.0, focal_loss_alpha=1.0, class_weights=None, label_as_inputs=False,
                 label_as_backbone_inputs=False, label_as_head_inputs=False, image_as_head_inputs=False, use_custom_logits=False,
                 logits_conv_postfix=None, logits_upsample_masks=None, resnet_multi_grids=[1, 2, 4], efficientnet_use_top=True,
                 dict_inputs_image_key="image", backbone_outputs_dict_key="endpoints", head_results_direct_output=False,
                 use_dict_outputs=False, **kwargs):
        super().__init__(num_class=num_class, input_norm_type=input_norm_type, custom_main_loss_fn=custom_main_loss_fn,
                         num_aux_loss=num_aux_loss, aux_loss_rate=aux_loss_rate, aux_metric_names=aux_metric_names, use_ohem=use_ohem,
                         ohem_thresh=ohem_thresh, use_focal_loss=use_focal_loss, focal_loss_gamma=focal_loss_gamma,
                         focal_loss_alpha=focal_loss_alpha, class_weights=class_weights, label_as_inputs=label_as_inputs,
                         custom_aux_loss_fns=custom_aux_loss_fns, **kwargs)
        self.backbone_name = backbone_name
        self.backbone_weights_path = backbone_weights_path
        self.output_stride = output_stride
        self.label_as_backbone_inputs = label_as_backbone_inputs
        self.label_as_head_inputs = label_as_head_inputs
        self.image_as_head_inputs = image_as_head_inputs
        self.use_custom_logits = use_custom_logits
        self.logits_upsample_masks = logits_upsample_masks
        self.head = None
        build_input_size = list(build_input_size)
        image_shape = (1, build_input_size[0], build_input_size[1], 3)
        self.backbone = get_backbone(self.backbone_name, custom_backbone_fn=backbone_custom_fn, output_stride=self.output_stride,
                                     weights_path=self.backbone_weights_path, return_endpoints=True, image_shape=image_shape,
                                     label_shape=None, resnet_multi_grids=resnet_multi_grids,
                                     efficientnet_use_top=efficientnet_use_top)
        if not self.use_custom_logits:
            logits_conv_name = "logits_conv"
            if logits_conv_postfix is not None:
                logits_conv_name = f"{logits_conv_name}_{logits_conv_postfix}"
            self.logits_conv = Conv2D(self.num_class, (1, 1), name=f"{self.name}/{logits_conv_name}")
            self.aux_logits_convs = self.build_aux_logits_conv(self.num_aux_loss, self.aux_metric_names)
        self.layers_for_multi_optimizers = None
        self.dict_inputs_image_key = dict_inputs_image_key
        self.backbone_outputs_dict_key = backbone_outputs_dict_key
        self.head_results_direct_output = head_results_direct_output
        self.use_dict_outputs = use_dict_outputs
        self._build_input_shape = image_shape

    def build_aux_logits_conv(self, num_aux_loss, aux_metric_names=None):
        convs = []
        for i in range(num_aux_loss):
            prefix = "aux" if aux_metric_names is None else aux_metric_names[i]
            convs.append(Conv2D(self.num_class, (1, 1), name=f"{self.name}/{prefix}_logits_conv_{i}"))
        return torch.nn.ModuleList(convs)

    def compute_backbone_results(self, backbone_inputs, training=None):
        return self.backbone(backbone_inputs, training=training)

    def compute_head_results(self, head_inputs, training=None):
        head_results = self.head(head_inputs, training=training)
        if isinstance(head_results, tuple):
            head_results = list(head_results)
        if not isinstance(head_results, list):
            head_results = [head_results]
        return head_results

    def compute_logits_results(self, logits_inputs):
        if not self.use_custom_logits:
            logits_list = [self.logits_conv(logits_inputs[0])]
            for i in range(len(self.aux_logits_convs)):
                logits_list += [self.aux_logits_convs[i](logits_inputs[i + 1])]
        else:
            logits_list = logits_inputs
        return logits_list

    def upsample_single_logits(self, logits, target_size):
        resize_method = "nearest" if logits.dtype is torch.int32 else "bilinear"
        return resize_image(logits, target_size, method=resize_method)

    def compute_logits_upsample(self, logits_list, inputs_size):
        if self.logits_upsample_masks is None:
            return [self.upsample_single_logits(l, inputs_size) for l in logits_list]
        assert len(self.logits_upsample_masks) == len(logits_list)
        y = []
        for i, logits in enumerate(logits_list):
            if self.logits_upsample_masks[i]:
                logits = self.upsample_single_logits(logits, inputs_size)
            y += [logits]
        return y

    def compute_final_results(self, logits_list):
        # the Keras-2 branch of the reference (:229-256) is the behavioural truth: list of float32 tensors
        return [l if isinstance(l, F.DeferredLogits) else
                (F.cast_to(l, torch.float32) if torch.is_tensor(l) else [F.cast_to(t, torch.float32) for t in l]) for l in logits_list]

    def call(self, inputs, training=None):
        return self._call_internal(inputs, training=training)

    def _call_internal(self, inputs, training=None):
        x = inputs
        label = None
        if self.label_as_inputs:
            x, label = x
        inputs_size = [int(x.shape[1]), int(x.shape[2])]
        backbone_inputs = x if not (self.label_as_inputs and self.label_as_backbone_inputs) else [x, label]
        endpoints = self.compute_backbone_results(backbone_inputs, training=training)
        head_inputs = endpoints
        if self.label_as_inputs and self.label_as_head_inputs:
            head_inputs = [endpoints, label]
        if self.image_as_head_inputs:
            head_inputs = [head_inputs, x] if not isinstance(head_inputs, list) or head_inputs is endpoints else head_inputs + [x]
        if self.head is None:
            raise ValueError("SegManaged.head is not set (the reference leaves the head to the user model, core_model_ext.py:91)")
        head_results = self.compute_head_results(head_inputs, training=training)
        if self.head_results_direct_output:
            return head_results
        logits_list = self.compute_logits_results(head_results)
        # fuse "bilinear upsample + cast to float32" into one pass (resize writes fp32 directly)
        # inside CoreTrain's step (F.defer_logits_upsample) the loss kernel does the upsample itself: hand over the low-resolution logits
        defer = bool(training) and F.deferring_logits_upsample()
        logits_list = [(F.DeferredLogits(l, inputs_size) if (defer and torch.is_tensor(l) and l.dim() == 4 and l.is_floating_point())
                        else F.resize_bilinear(l, inputs_size, out_dtype=torch.float32))
                       if (self.logits_upsample_masks is None or self.logits_upsample_masks[i]) else l
                       for i, l in enumerate(logits_list)]
        logits_list = self.compute_final_results(logits_list)
        if self.use_dict_outputs:
            return {self._index_to_output_key(i): l for i, l in enumerate(logits_list)}
        return logits_list

    def build_with_dummy(self):
        """build every lazily-built layer (head, logits convs) by shape propagation"""
        from .. import nn

        with nn.dry_run_scope():
            self(torch.empty(self._build_input_shape, dtype=torch.float32, device=nn.device()), training=False)
        return self

    def multi_optimizers_layers(self):
        return self.layers_for_multi_optimizers

    def on_epoch_end(self, epoch, logs={}):
        for part in (self.head, self.backbone):
            fn = getattr(part, "on_epoch_end", None)
            if callable(fn):
                fn(epoch, logs)
