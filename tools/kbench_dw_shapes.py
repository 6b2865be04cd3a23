"""the depthwise 7 x 7 forward / data-gradient / weight-gradient launches of the four ConvNeXt-T stages (16 images), a few launches each:
run under tools/ktrace.sh (kernel-only times) or tools/ktraffic.sh (HBM-side bytes per launch)"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from iseg_amd import kernels as K
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
flush = os.environ.get("KB_FLUSH", "1") == "1"      # 0: back-to-back launches on cache-warm tensors
for (S, C) in [(128, 96), (64, 192), (32, 384), (16, 768)]:
    x = torch.randn(16, S, S, C, device="cuda").to(torch.bfloat16)
    r = torch.randn(16, S, S, C, device="cuda").to(torch.bfloat16)
    wd = torch.randn(49, C, device="cuda") / 7; bd = torch.randn(C, device="cuda")
    dwg = torch.zeros(49, C, device="cuda"); dbg = torch.zeros(C, device="cuda")
    big = torch.empty(512 << 20, dtype=torch.uint8, device="cuda") if flush else None
    for _ in range(n):
        if flush: big.zero_()      # (cold caches between launches: 512 MiB through L2 / Infinity Cache)
        K.dwconv2d(x, wd, bd, 7, 1, 3, 3)
        if flush: big.zero_()
        K.dwconv2d(x, wd, None, 7, 1, 3, 3, flip=True, add=r)
        if flush: big.zero_()
        K.dwconv2d_bwd_weight(x, r, dwg, dbg, 7, 1, 3, 3)
    torch.cuda.synchronize()
