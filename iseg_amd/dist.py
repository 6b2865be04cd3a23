"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm) / gloo on CPU.

Replaces distribution/distribution_utils.py:75-95,158-169 (tf.distribute.MirroredStrategy + ReplicaContext.all_reduce):
  * SyncBN statistics: one packed [2C+1] fp32 all-reduce per layer in forward, one [2C] in backward (functional.py);
  * gradients: the flat fp32 gradient buffer is cut into contiguous buckets; a bucket's all-reduce(sum) is launched
    asynchronously as soon as every parameter in it has its gradient enqueued, so RCCL traffic over xGMI overlaps the
    rest of the backward pass.  xGMI is a point-to-point mesh (7 links/GPU), so few large messages are used.
"""
import os

import torch
import torch.distributed as td


def is_initialized():
    return td.is_available() and td.is_initialized()


def world_size():
    return td.get_world_size() if is_initialized() else 1


def rank():
    return td.get_rank() if is_initialized() else 0


def local_rank():
    return int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT)."""
    if is_initialized():
        return
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1:
        return
    if backend is None:
        backend = os.environ.get("ISEG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank())
        from . import kernels

        kernels._DEVICE_INDEX[0] = None      # the launch path caches the device index (kernels.stream)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    td.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=ws)


def all_reduce_sum(t, async_op=False):
    if world_size() > 1:
        return td.all_reduce(t, op=td.ReduceOp.SUM, async_op=async_op)
    return None


def broadcast(t, src=0):
    if world_size() > 1:
        td.broadcast(t, src)


def barrier():
    if world_size() > 1:
        td.barrier()


class GradReducer:
    """Bucketed, backward-overlapped all-reduce(sum) over the flat gradient buffer of a ParamStore."""

    def __init__(self, store, bucket_bytes=48 << 20):
        self.store = store
        self.buckets = []      # (lo, hi, n_params)
        self.bucket_of = {}    # id(param) -> bucket index
        lo, cnt, acc = None, 0, 0
        for p, off, n in store.segments:
            if lo is None:
                lo = off
            self.bucket_of[id(p)] = len(self.buckets)
            cnt += 1
            acc += store.padded(n) * 4
            if acc >= bucket_bytes:
                self.buckets.append((lo, off + store.padded(n), cnt))
                lo, cnt, acc = None, 0, 0
        if cnt:
            last = store.segments[-1]
            self.buckets.append((lo, last[1] + store.padded(last[2]), cnt))
        self.reset()

    def reset(self):
        self.pending = [b[2] for b in self.buckets]
        self.handles = []

    def ready(self, *params):
        if world_size() <= 1:
            return
        for p in params:
            if p is None:
                continue
            b = self.bucket_of.get(id(p))
            if b is None:
                continue
            self.pending[b] -= 1
            if self.pending[b] == 0:
                lo, hi, _ = self.buckets[b]
                self.handles.append(all_reduce_sum(self.store.flat_g[lo:hi], async_op=True))

    def finish(self):
        """launch whatever was not triggered by ready() and make the compute stream wait for all buckets"""
        if world_size() > 1:
            for b, left in enumerate(self.pending):
                if left > 0:
                    lo, hi, _ = self.buckets[b]
                    self.handles.append(all_reduce_sum(self.store.flat_g[lo:hi], async_op=True))
            for h in self.handles:
                if h is not None:
                    h.wait()
        self.reset()


_ACTIVE_REDUCER = [None]


def set_active_reducer(r):
    _ACTIVE_REDUCER[0] = r


def grads_ready(*params):
    r = _ACTIVE_REDUCER[0]
    if r is not None:
        r.ready(*params)
