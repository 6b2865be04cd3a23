#!/usr/bin/env python3
"""Throughput of the other BASELINE configurations on one GPU (bf16, synthetic data, full train step through TrainableModel, or the
sliding-window inference driver for cfg4).  One JSON line per configuration.  Not the contract benchmark (that is bench.py / cfg2).

  python tools/bench_configs.py [cfg1 cfg3 cfg4 cfg4_infer cfg5 v2 hrnet] [--steps K] [--warmup W] [--batch B]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

CONFIGS = {
    # name: (factory, size, default batch, training, description)
    "cfg1": ("resnet50_aspp", 256, 16, True, "ResNet-50 + ASPP 256x256"),
    "cfg3": ("swin_tiny_fpn", 512, 16, True, "Swin-T + FPN 512x512"),
    "cfg4": ("vit_base_simple_decoder", 512, 8, True, "ViT-B/16 + SimpleDecoder 512x512 (train step)"),
    "cfg4_infer": ("vit_base_simple_decoder", 640, 1, False, "ViT-B/16 + SimpleDecoder 640x640, sliding window 512"),
    "cfg5": ("intern_image_base_aspp", 512, 8, True, "InternImage-B + ASPP 512x512"),
    "hrnet": ("hrnet_w32_aspp", 512, 8, True, "HRNet-W32 + ASPP 512x512 (not a BASELINE configuration)"),
    "v2": ("convnext_v2_tiny_aspp", 512, 16, True, "ConvNeXt-V2-T + ASPP 512x512 (not a BASELINE configuration: the next backbone family)"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*", default=[n for n in CONFIGS if n.startswith("cfg")])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--each-step", action="store_true", help="also print every step's own synchronised time, warm-up included (looks for transients)")
    ap.add_argument("--graph-step", action="store_true", help="replay the train step from one HIP graph (iseg_amd/graphs.py GraphedTrainStep)")
    args = ap.parse_args()
    from iseg_amd import heads
    from iseg_amd.core_env import common_env_setup
    from iseg_amd.core_inference import inference_with_sliding_window
    from iseg_amd.core_optimizer import get_optimizer
    from iseg_amd.core_train import CoreTrain
    from iseg_amd.data import synthetic_batch
    from iseg_amd.modelhelper import model_common_setup

    strategy = common_env_setup(use_one_device_strategy=True, mixed_precision=True, random_seed=0)
    for name in args.names:
        factory, size, batch, training, desc = CONFIGS[name]
        batch = args.batch or batch
        build = (size, size) if training else (512, 512)
        model = getattr(heads, factory)(build_input_size=build)
        helper = model_common_setup(model, restore_checkpoint=False)
        x, y = synthetic_batch(batch, size, size, seed=7)
        x, y = x.cuda(), y.cuda()
        if training:
            helper.set_optimizer(get_optimizer(strategy, initial_lr=1e-4, end_lr=0.0, epoch_steps=1000, train_epoch=30, optimizer="adamw",
                                               adamw_weight_decay=0.05))
            trainer = CoreTrain(helper, None).create_trainable_model(21, ignore_label=255, batch_size=batch)

            if args.graph_step:
                from iseg_amd.graphs import GraphedTrainStep

                graphed = GraphedTrainStep(trainer, warmup=2)

                def step():
                    return graphed(x, y)
            else:
                def step():
                    return trainer.train_step(x, y)
        elif args.graph_step:
            from iseg_amd.graphs import graphed_inference

            ginf = graphed_inference(model, (512, 512))      # the whole sliding-window call replayed from one HIP graph per input shape

            def step():
                return ginf(x)
        else:
            def step():
                with torch.no_grad():
                    return inference_with_sliding_window(x, model, training=False, windows_size=(512, 512))
        each = []
        # a capture leaves the GPU idle for ~0.4 s and the first replays behind it run slow (13.3, 12.7, 12.5, 12.4, 12.3 ms on cfg3: clocks coming back up;
        # profiles/r06_cfg3_replay_vs_eager.txt) -- how long that lasts differs from box to box, so the replayed form gets twelve more untimed steps
        for _ in range(args.warmup + (12 if args.graph_step else 0)):
            t1 = time.perf_counter()
            step()
            if args.each_step:
                torch.cuda.synchronize()
                each.append(round((time.perf_counter() - t1) * 1e3, 2))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        host = 0.0
        for _ in range(args.steps):
            t1 = time.perf_counter()
            out = step()
            host += time.perf_counter() - t1
            if args.each_step:
                torch.cuda.synchronize()
                each.append(round((time.perf_counter() - t1) * 1e3, 2))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        rec = {"config": name, "workload": desc, "batch": batch, "dtype": "bf16", "ms_per_step": round(dt * 1e3, 3),
               "images_per_sec": round(batch / dt, 2), "params_M": round(sum(p.numel() for p in model.parameters()) / 1e6, 2)}
        rec["step"] = "hip graph replay" if args.graph_step else "eager"
        rec["host_enqueue_ms_per_step"] = round(host / args.steps * 1e3, 3)      # wall time the host spent inside step() (no synchronisation)
        if training:
            rec["loss"] = round(float(out[0]), 4)
        if args.each_step:
            rec["each_step_ms"] = each
        print(json.dumps(rec), flush=True)
        del model, helper
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
