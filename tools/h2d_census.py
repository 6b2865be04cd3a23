#!/usr/bin/env python3
"""Where the small host->device and device->device copies of one flagship train step come from (rocprof shows them as __amd_rocclr_copyBuffer): wraps the torch entry
points that can move host data to the device and prints the call sites with counts.   python tools/h2d_census.py [cfg]"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

sites = collections.Counter()
on = [False]


def site():
    for f in reversed(traceback.extract_stack()[:-2]):
        if "iseg_amd" in f.filename or f.filename.endswith("bench.py"):
            return f"{os.path.relpath(f.filename)}:{f.lineno} {f.name}"
    return "?"


def wrap_method(cls, name, is_h2d):
    orig = getattr(cls, name)

    def w(self, *a, **k):
        if on[0] and is_h2d(self, a, k):
            sites[f"Tensor.{name} <- {site()}"] += 1
        return orig(self, *a, **k)

    setattr(cls, name, w)


def dev_of(a, k):
    d = k.get("device")
    for x in a:
        if isinstance(x, (torch.device, str)):
            d = x
        if torch.is_tensor(x):
            d = x.device
    return str(d) if d is not None else ""


wrap_method(torch.Tensor, "to", lambda s, a, k: not s.is_cuda and "cuda" in dev_of(a, k))
wrap_method(torch.Tensor, "cuda", lambda s, a, k: not s.is_cuda)
wrap_method(torch.Tensor, "copy_", lambda s, a, k: s.is_cuda and len(a) > 0 and torch.is_tensor(a[0]) and not a[0].is_cuda)
wrap_method(torch.Tensor, "copy_", lambda s, a, k: s.is_cuda and len(a) > 0 and torch.is_tensor(a[0]) and a[0].is_cuda)      # device -> device (rocclr blit when contiguous)
wrap_method(torch.Tensor, "clone", lambda s, a, k: s.is_cuda)
wrap_method(torch.Tensor, "contiguous", lambda s, a, k: s.is_cuda and not s.is_contiguous())
wrap_method(torch.Tensor, "float", lambda s, a, k: s.is_cuda and s.dtype != torch.float32)
wrap_method(torch.Tensor, "fill_", lambda s, a, k: s.is_cuda)
wrap_method(torch.Tensor, "zero_", lambda s, a, k: s.is_cuda)
for fname in ("tensor", "as_tensor", "full", "zeros", "ones", "arange", "scalar_tensor"):
    orig = getattr(torch, fname)

    def mk(orig=orig, fname=fname):
        def w(*a, **k):
            if on[0] and "cuda" in str(k.get("device", "")):
                sites[f"torch.{fname}(device=cuda) <- {site()}"] += 1
            return orig(*a, **k)
        return w

    setattr(torch, fname, mk())

sys.argv = [sys.argv[0]]
args = bench.parse()
from iseg_amd.data import synthetic_batch  # noqa: E402

strategy, model, trainer = bench.build_trainer(args)
x, y = synthetic_batch(args.batch, args.size, args.size, seed=100)
x, y = x.cuda(), y.cuda()
for _ in range(4):
    trainer.train_step(x, y)
torch.cuda.synchronize()
on[0] = True
steps = 3
for _ in range(steps):
    trainer.train_step(x, y)
torch.cuda.synchronize()
on[0] = False
for k, v in sites.most_common():
    print(f"{v / steps:6.1f} / step  {k}")
