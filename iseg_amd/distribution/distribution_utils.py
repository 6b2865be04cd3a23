"""distribution/distribution_utils.py of the reference (:59-169) re-expressed for one-process-per-GPU data parallelism:
the Strategy object stands in for tf.distribute.{OneDevice,Mirrored}Strategy (`with strategy.scope():`,
`num_replicas_in_sync`) and owns the RCCL process group; all_reduce_values is ReplicaContext.all_reduce(SUM)."""
import contextlib
import os

import torch

from .. import dist
from .. import nn


class Strategy:
    def __init__(self, one_device=False):
        self.one_device = one_device
        if not one_device:
            dist.init()
        self.rank = dist.rank()
        self.world_size = 1 if one_device else dist.world_size()
        if torch.cuda.is_available():
            torch.cuda.set_device(dist.local_rank())
            nn.set_device(torch.device("cuda", dist.local_rank()))

    @property
    def num_replicas_in_sync(self):
        return self.world_size

    @contextlib.contextmanager
    def scope(self):
        yield self


def list_gpus():
    return [f"cuda:{i}" for i in range(torch.cuda.device_count())]


def get_gpu_counts():
    return torch.cuda.device_count()


def set_gpu_memory_growth(growth=False):
    return None     # HIP allocations grow on demand through torch's caching allocator


def build_one_device_strategy(device=None):
    return Strategy(one_device=True)


def build_mirrored_strategy(dist_devices=None):
    return Strategy(one_device=False)


def get_distribution_strategy(gpu_memory_growth=True, cuda_visible_devices=None, use_tpu=False, tpu_name=None,
                              use_one_device_strategy=False):
    if use_tpu:
        raise ValueError("TPU strategies are outside the MI355X port")
    if cuda_visible_devices is not None:
        os.environ.setdefault("HIP_VISIBLE_DEVICES", str(cuda_visible_devices))
    if use_one_device_strategy:
        return build_one_device_strategy()
    return build_mirrored_strategy()


def all_reduce_values(vars, reduce_op="sum"):
    if reduce_op not in ("sum", "SUM"):
        raise ValueError("only SUM all-reduce is used by the reference")
    if isinstance(vars, (list, tuple)):
        for v in vars:
            dist.all_reduce_sum(v)
        return vars
    dist.all_reduce_sum(vars)
    return vars
