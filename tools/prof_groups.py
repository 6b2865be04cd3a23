#!/usr/bin/env python3
"""Summarise a `rocprofv3 --kernel-trace` run of bench.py: time per step by kernel family and by (kernel, grid) group.
usage: python3 tools/prof_groups.py <rocprof output dir> <steps traced (steps + warmup)> [rows] [out.md]"""
import collections
import csv
import glob
import re
import sys

FAMILIES = [
    ("fused ConvNeXt MLP (fwd / bwd chain / prep)", r"convnext_mlp"),
    ("GEMM (register-staged)", r"gemm_bf16_kernel|gemm_f32_kernel"),
    ("GEMM (LDS-DMA)", r"gemm_bf16_dma_kernel|gemm_bf16_dma_tn_kernel|gemm_bf16_dma_tn_pair_kernel"),
    ("implicit-GEMM conv", r"igemm"),
    ("split-K / partial reductions", r"splitk_reduce|reduce_rows"),
    ("depthwise conv fwd / bwd-data", r"dwconv_fwd|dwconv7_mfma"),
    ("depthwise conv weight grad", r"dwconv_bwd_weight"),
    ("LayerNorm", r"layernorm"),
    ("BatchNorm", r"bn_"),
    ("im2col / col2im", r"im2col|col2im"),
    ("loss / metric / resize tail", r"softmax_ce|argmax|resize|upsample_ce"),
    ("optimizer", r"adamw|sgd_kernel"),
    ("layer scale / column sums / row scale", r"layerscale|colsum|rowscale|scale_cols|mul_colsum"),
    ("attention", r"attn|softmax_rows|relpos|gather_rows"),
    ("ATen / runtime (not ours)", r"at::native|rocclr|elementwise_kernel"),
]


def main():
    d, steps = sys.argv[1], int(sys.argv[2])
    nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    out = open(sys.argv[4], "w") if len(sys.argv) > 4 else sys.stdout
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    groups, fam = collections.defaultdict(list), collections.defaultdict(float)
    launches = 0
    for r in csv.DictReader(open(f)):
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        name = r["Kernel_Name"]
        groups[(name[:110], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])].append(us)
        launches += 1
        for label, pat in FAMILIES:
            if re.search(pat, name):
                fam[label] += us
                break
        else:
            fam["other"] += us
    tot = sum(sum(v) for v in groups.values())
    print(f"kernel time {tot / steps / 1e3:.3f} ms/step, {launches / steps:.0f} launches/step ({steps} steps traced)\n", file=out)
    print("| ms/step | share | family |\n|---|---|---|", file=out)
    for label, us in sorted(fam.items(), key=lambda kv: -kv[1]):
        print(f"| {us / steps / 1e3:.3f} | {100 * us / tot:.1f} % | {label} |", file=out)
    print("\n| ms/step | launches/step | avg us | kernel | grid x | grid y | block |\n|---|---|---|---|---|---|---|", file=out)
    for k, v in sorted(groups.items(), key=lambda kv: -sum(kv[1]))[:nrows]:
        print(f"| {sum(v) / steps / 1e3:.3f} | {len(v) / steps:.1f} | {sum(v) / len(v):.1f} | `{k[0]}` | {k[1]} | {k[2]} | {k[3]} |", file=out)


if __name__ == "__main__":
    main()
