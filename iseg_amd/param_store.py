"""Flat parameter / gradient / optimizer-state storage.

Every parameter of a model becomes a view into ONE fp32 buffer (each tensor padded to a multiple of 256 elements), its
.grad a view into ONE fp32 gradient buffer, and -- under mixed precision -- its bf16 compute copy a view into ONE bf16
buffer.  That makes the optimizer a single kernel launch (csrc/optim.hip), the gradient all-reduce a handful of large
contiguous RCCL messages (dist.GradReducer) and "zero the gradients" one memset.
"""
import torch

from . import kernels as K
from . import nn

ALIGN = 256


class ParamStore:
    def __init__(self, params):
        seen, plist = set(), []
        for p in params:
            if id(p) not in seen:
                seen.add(id(p))
                plist.append(p)
        if not plist:
            raise ValueError("ParamStore: no parameters")
        dev = plist[0].device
        self.device = dev
        self.segments = []   # (param, offset, numel)
        off = 0
        for p in plist:
            n = p.numel()
            self.segments.append((p, off, n))
            off += self.padded(n)
        self.total = off
        self.flat_w = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_bf16 = torch.zeros(off, dtype=torch.bfloat16, device=dev)
        seg_of_block = torch.full((off // ALIGN,), -1, dtype=torch.int32)
        for i, (p, o, n) in enumerate(self.segments):
            self.flat_w[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat_w[o:o + n].view(p.shape)
            p.grad = self.flat_g[o:o + n].view(p.shape)
            p.iseg_compute = self.flat_bf16[o:o + n].view(p.shape)
            seg_of_block[o // ALIGN:(o + self.padded(n)) // ALIGN] = i
        self.seg_of_block = seg_of_block.to(dev)
        self.nblocks = off // ALIGN
        self.sync_shadow()

    @staticmethod
    def padded(n):
        return (n + ALIGN - 1) // ALIGN * ALIGN

    @property
    def params(self):
        return [s[0] for s in self.segments]

    def sync_shadow(self):
        """refresh the bf16 compute copies from the fp32 masters (one cast kernel)"""
        if self.flat_w.is_cuda:
            K.cast(self.flat_w, torch.bfloat16, out=self.flat_bf16)
        else:
            self.flat_bf16.copy_(self.flat_w)   # host-side layout only (no GPU): never on the compute path
        nn.weights_changed()

    def zero_grad(self):
        if self.flat_g.is_cuda:
            K.fill_f32(self.flat_g, 0.0)      # (torch's .zero_() is an ATen fill kernel)
        else:
            self.flat_g.zero_()

    def broadcast_from_rank0(self):
        from . import dist

        dist.broadcast(self.flat_w, 0)
        self.sync_shadow()
