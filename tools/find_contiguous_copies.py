#!/usr/bin/env python3
"""Which call sites materialise a copy through Tensor.contiguous() during one training step of a tools/bench_configs.py configuration
(each is a runtime blit kernel on the step).   python tools/find_contiguous_copies.py hrnet"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools.host_time import build_config  # noqa: E402


def main():
    trainer, x, y = build_config(sys.argv[1] if len(sys.argv) > 1 else "hrnet")
    for _ in range(2):
        trainer.train_step(x, y)
    sites = collections.Counter()
    real = torch.Tensor.contiguous

    def spy(self, *a, **k):
        if not self.is_contiguous():
            st = traceback.extract_stack(limit=5)[:-1]
            sites[" <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(st[-3:]))] += self.numel() * self.element_size()
        return real(self, *a, **k)

    torch.Tensor.contiguous = spy
    try:
        trainer.train_step(x, y)
    finally:
        torch.Tensor.contiguous = real
    for site, nbytes in sites.most_common(12):
        print(f"{nbytes / 2 ** 20:9.1f} MiB  {site}")


if __name__ == "__main__":
    main()
