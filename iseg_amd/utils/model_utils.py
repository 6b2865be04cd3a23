"""utils/model_utils.py of the reference (:20-88): create_compiled_model -- gather losses / loss weights / metrics from the
model, exclude the no-weight-decay variables, "compile" (here: bind everything into a TrainableModel), set the optimizer
iteration counter for LR-schedule resume."""
from ..core_model import SegFoundation
from ..losses.catecrossentropy_ignore_label import catecrossentropy_ignore_label_loss
from ..metrics.mean_iou import MeanIOU
from ..metrics.seg_metric_wrapper import SegMetricWrapper
from ..trainer import TrainableModel
from .keras_ops import capture_func
from .train_utils import exclude_no_weight_decay_layers_in_optimizer


def create_compiled_model(model: SegFoundation, num_class, ignore_label=255, class_weights=None, batch_size=1, epoch_steps=1000,
                          initial_epoch=0, jit_compile=None, optimizer=None):
    assert isinstance(model, SegFoundation), "Current only support SegFoundation based model"
    losses_func = getattr(model, "custom_losses", None)
    if losses_func is None or not callable(losses_func):
        losses_func = catecrossentropy_ignore_label_loss
    losses = losses_func(num_class=num_class, ignore_label=ignore_label, class_weights=class_weights, batch_size=batch_size,
                         reduction=False)
    losses_weights = None
    losses_weights_func = capture_func(model, "custom_losses_weights")
    if losses_weights_func is not None:
        losses_weights = losses_weights_func()
    metrics_func = getattr(model, "custom_metrics", None)
    if metrics_func is None or not callable(metrics_func):
        metrics_func = _get_default_metrics
    metrics = metrics_func(num_class, ignore_label)
    if optimizer is not None:
        exclude_no_weight_decay_layers_in_optimizer(optimizer=optimizer, model=model, print_excluded_list=False)
    compiled = TrainableModel(model, optimizer=optimizer, loss=losses, loss_weights=losses_weights, metrics=metrics,
                              jit_compile=jit_compile)
    if initial_epoch != -1 and optimizer is not None:
        for opt in (optimizer if isinstance(optimizer, list) else [optimizer]):
            opt.iterations = epoch_steps * initial_epoch
    return compiled


def _get_default_metrics(num_class, ignore_label):
    iou_metrics = MeanIOU(num_class)
    iou_metrics = SegMetricWrapper(iou_metrics, num_class=num_class, ignore_label=ignore_label, name="IOU")
    return [iou_metrics]
