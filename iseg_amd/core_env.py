"""core_env.py of the reference (:16-87): common_env_setup -> strategy; seeds, determinism, mixed precision."""
from .distribution.distribution_utils import get_distribution_strategy
from .utils.common import enable_mixed_precision, set_random_seed


def common_env_setup(run_eagerly=False, gpu_memory_growth=True, cuda_visible_devices=None, use_mesh=False,
                     use_one_device_strategy=False, tpu_name=None, random_seed=0, mixed_precision=True, use_deterministic=True,
                     num_op_parallelism_threads=-1, numpy_behavior=False, soft_device_placement=False):
    set_random_seed(random_seed)
    use_tpu = tpu_name is not None
    print(f"Using TPU: {use_tpu}")
    # use_deterministic (reference: enable_op_determinism(), core_env.py:39-48, default True): honoured unconditionally -- no kernel of the step
    # uses a floating-point atomic (every cross-lane / cross-workgroup sum runs in a fixed order; integer atomics commute), so two runs from the
    # same state give the same bits (tests/test_graph_train_gpu.py).  False changes nothing: there is no faster non-deterministic variant to opt into.
    print(f"use_deterministic = {use_deterministic} (the MI355X path is bit-reproducible either way)")
    strategy = get_distribution_strategy(gpu_memory_growth=gpu_memory_growth, cuda_visible_devices=cuda_visible_devices,
                                         use_tpu=use_tpu, tpu_name=tpu_name, use_one_device_strategy=use_one_device_strategy)
    if mixed_precision:
        enable_mixed_precision(use_tpu=False)
    else:
        import torch

        from . import nn

        nn.set_compute_dtype(torch.float32)
    return strategy


def common_env_clean(strategy):
    return None
