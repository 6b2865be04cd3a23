"""Minimal stand-in for the tf.data pipeline contract CoreTrain relies on (core_train.py:155-167 of the reference):
an iterable of (image[H,W,3] float32, label[H,W] int32) supporting map / shuffle / repeat / batch / prefetch, plus the
synthetic generator of SURVEY.md section 8(d).  Host-side only; batches are staged to the GPU through pinned memory."""
import queue
import random
import threading

import numpy as np
import torch


class Dataset:
    def __init__(self, gen_fn):
        self._gen_fn = gen_fn

    def __iter__(self):
        return iter(self._gen_fn())

    @staticmethod
    def from_tensors_list(items):
        return Dataset(lambda: iter(items))

    @staticmethod
    def from_generator(fn):
        return Dataset(fn)

    def map(self, fn, num_parallel_calls=None):
        src = self

        def gen():
            for item in src:
                yield fn(*item) if isinstance(item, tuple) else fn(item)

        return Dataset(gen)

    def shuffle(self, buffer_size, seed=0, reshuffle_each_iteration=True):
        """tf.data semantics: every new iteration (each epoch of a following .repeat()) draws a different order"""
        src = self
        epoch = [0]

        def gen():
            rng = random.Random(seed * 1000003 + epoch[0])
            if reshuffle_each_iteration:
                epoch[0] += 1
            buf = []
            for item in src:
                buf.append(item)
                if len(buf) >= buffer_size:
                    yield buf.pop(rng.randrange(len(buf)))
            while buf:
                yield buf.pop(rng.randrange(len(buf)))

        return Dataset(gen)

    def repeat(self, count=None):
        src = self

        def gen():
            n = 0
            while count is None or n < count:
                empty = True
                for item in src:
                    empty = False
                    yield item
                if empty:
                    return
                n += 1

        return Dataset(gen)

    def batch(self, batch_size, drop_remainder=False):
        src = self

        def stack(items):
            first = items[0]
            if isinstance(first, tuple):
                return tuple(stack([it[i] for it in items]) for i in range(len(first)))
            return torch.stack([torch.as_tensor(t) for t in items])

        def gen():
            cur = []
            for item in src:
                cur.append(item)
                if len(cur) == batch_size:
                    yield stack(cur)
                    cur = []
            if cur and not drop_remainder:
                yield stack(cur)

        return Dataset(gen)

    def shard(self, num_shards, index):
        src = self

        def gen():
            for i, item in enumerate(src):
                if i % num_shards == index:
                    yield item

        return Dataset(gen)

    def prefetch(self, buffer_size=2, device=None):
        src = self
        if buffer_size is None or buffer_size < 1:
            buffer_size = 2

        def to_dev(t):
            if device is None or not isinstance(t, torch.Tensor):
                return t
            if device.type == "cuda":
                return t.pin_memory().to(device, non_blocking=True)
            return t.to(device)

        def gen():
            q = queue.Queue(maxsize=buffer_size)
            stop = object()

            def worker():
                try:
                    for item in src:
                        q.put(item)
                finally:
                    q.put(stop)

            th = threading.Thread(target=worker, daemon=True)
            th.start()
            while True:
                item = q.get()
                if item is stop:
                    return
                yield tuple(to_dev(t) for t in item) if isinstance(item, tuple) else to_dev(item)

        return Dataset(gen)


def synthetic_batch(batch, height, width, num_class=21, ignore_label=255, seed=0, block=1):
    """images ~ U(-1,1) (seed), labels ~ U{0..num_class-1} with 10% of the pixels = ignore_label (seed+1);
    block>1 gives block-constant label maps."""
    rng = np.random.default_rng(seed)
    img = rng.uniform(-1.0, 1.0, size=(batch, height, width, 3)).astype(np.float32)
    rng_l = np.random.default_rng(seed + 1)
    if block > 1:
        lab = rng_l.integers(0, num_class, size=(batch, -(-height // block), -(-width // block)), dtype=np.int32)
        lab = np.repeat(np.repeat(lab, block, axis=1), block, axis=2)[:, :height, :width]
    else:
        lab = rng_l.integers(0, num_class, size=(batch, height, width), dtype=np.int32)
    mask = rng_l.random(size=(batch, height, width)) < 0.1
    lab = np.where(mask, np.int32(ignore_label), lab).astype(np.int32)
    return torch.from_numpy(img), torch.from_numpy(np.ascontiguousarray(lab))


def synthetic_dataset(num_samples, height, width, num_class=21, ignore_label=255, seed=0):
    img, lab = synthetic_batch(num_samples, height, width, num_class, ignore_label, seed)
    return Dataset.from_tensors_list([(img[i], lab[i]) for i in range(num_samples)])
