import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # the CPU oracle at the container's real core budget: torch would start 128 threads on a 16-core cgroup quota and run 8 x slower
    from oracle import host_threads

    host_threads.apply()


# Collection order (the driver runs `pytest -x`): the oracle / golden PARITY suites first -- kernels, fused blocks, optimizers, whole models --
# then the host-logic and feature suites in their usual order, and the self-comparison / property tests (graph replay vs eager, run-to-run
# identity, launcher refusals) LAST, so a property test can never hide the parity suite behind a first failure.
_PARITY_FIRST = ["test_oracle_known_answers", "test_golden_cpu", "test_golden_gpu", "test_kernels_gpu", "test_mlp_fused_gpu", "test_blocks_gpu",
                 "test_optimizer_gpu", "test_model_gpu", "test_configs_gpu", "test_model_builder_gpu", "test_misc_gpu", "test_upsample_ce_gpu",
                 "test_step_fusions_gpu", "test_attention_gpu", "test_conv_igemm_gpu", "test_dcnv3_gpu", "test_resnet_gpu", "test_focal_gpu",
                 "test_weights_import_gpu", "test_input_pipeline_gpu"]
_PROPERTY_LAST_FILES = ["test_graph_train_gpu", "test_determinism_gpu", "test_bench_launch"]
_PROPERTY_LAST_NAMES = ["bit_identical", "refuses", "reproducib", "run_to_run", "graphed", "graph_replay"]


def _order_key(item):
    path = item.nodeid.split("::")[0]
    stem = os.path.splitext(os.path.basename(path))[0]
    name = item.nodeid.lower()
    if stem in _PROPERTY_LAST_FILES or any(k in name.split("::", 1)[-1] for k in _PROPERTY_LAST_NAMES):
        return (2, 0)
    if stem in _PARITY_FIRST:
        return (0, _PARITY_FIRST.index(stem))
    return (1, 0)


def pytest_collection_modifyitems(config, items):
    items.sort(key=_order_key)      # (stable: the order inside a file, and of the files of one class, is kept)


@pytest.fixture(scope="session")
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
