"""evaluations/evaluation.py of the reference (:19-124): the evaluation loop (multi-scale + flip inference, running loss, mean IoU over a
dataset with a ragged last batch) against the oracle's multi-scale inference + ignore-label loss + confusion matrix."""
import pytest
import torch

from oracle import models as OM
from oracle import tf_ops as O
from tests.util_models import randomize_parameters

pytestmark = pytest.mark.gpu


def test_evaluate_matches_oracle_miou_and_loss(cuda):
    from iseg_amd import nn
    from iseg_amd.data import synthetic_dataset
    from iseg_amd.distribution.distribution_utils import Strategy
    from iseg_amd.evaluations import evaluate
    from iseg_amd.heads import convnext_tiny_aspp
    from iseg_amd.param_store import ParamStore

    nn.set_compute_dtype(torch.float32)
    nn.set_device("cuda:0")
    model = convnext_tiny_aspp(build_input_size=(64, 64), drop_path_rate=0.0, dropout_rate=0.0, layer_scale_init_value=1.0)
    model._iseg_store = ParamStore(list(model.parameters()))
    randomize_parameters(model, 2)
    data = synthetic_dataset(3, 64, 64, seed=11)      # batch 2 -> batches of 2, 1 (drop_remainder=False)
    scales = [0.75, 1.0]      # (every scale x flip is one more fp64 oracle forward per image on the host)
    miou = evaluate(Strategy(one_device=True), model, data, batch_size=2, num_class=21, ignore_label=255, scale_rates=scales, flip=True,
                    val_image_count=3, verbose=0)
    w = OM.export_weights(model)
    cm = torch.zeros(21, 21, dtype=torch.float64)
    losses = []
    for img, lab in data:
        logits = OM.multi_scale_inference(lambda t: OM.convnext_aspp_forward(w, t, training=False)["logits"], img[None].double(), tuple(scales), True)
        losses.append(O.softmax_ce_ignore(lab[None], logits, 21, 255))
        cm += O.confusion_matrix(lab[None], O.argmax_first(logits), 21, 255)
    _, want = O.per_class_iou(cm)
    want = float(want)
    assert abs(float(miou) - want) < 1e-6, (float(miou), want)
    want_loss = float(torch.cat([l.reshape(-1) for l in losses]).mean())
    assert abs(evaluate.last_mean_loss - want_loss) < 1e-4 * max(1.0, abs(want_loss))
    with pytest.raises(ValueError):
        evaluate(Strategy(one_device=True), object(), data, 2, 21)
