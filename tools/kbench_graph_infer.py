#!/usr/bin/env python3
"""cfg4's workload (ViT-B/16 + SimpleDecoder, one 640 x 640 image, sliding window 512) eager and as a replayed HIP graph."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    from iseg_amd import heads
    from iseg_amd.core_env import common_env_setup
    from iseg_amd.core_inference import inference_with_sliding_window
    from iseg_amd.data import synthetic_batch
    from iseg_amd.graphs import graphed_inference
    from iseg_amd.modelhelper import model_common_setup

    common_env_setup(use_one_device_strategy=True, mixed_precision=True, random_seed=0)
    name = sys.argv[1] if len(sys.argv) > 1 else "vit_base_simple_decoder"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 640
    model = getattr(heads, name)(build_input_size=(512, 512))
    model_common_setup(model, restore_checkpoint=False)
    x, _ = synthetic_batch(1, size, size, seed=7)
    x = x.cuda()

    def eager(v):
        with torch.no_grad():
            return inference_with_sliding_window(v, model, training=False, windows_size=(512, 512))

    g = graphed_inference(model, (512, 512))
    ref = eager(x).clone()
    for _ in range(4):
        out = g(x)
    torch.cuda.synchronize()
    print("max |graph - eager|:", (out - ref).abs().max().item())
    x2 = torch.roll(x, 17, dims=2)
    assert torch.equal(g(x2), eager(x2)), "a new input through the captured graph differs from the eager result"
    for fn, label in ((eager, "eager"), (g, "graph")):
        for _ in range(3):
            fn(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            fn(x)
        torch.cuda.synchronize()
        print(f"{label}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per {size}x{size} image")


if __name__ == "__main__":
    main()
