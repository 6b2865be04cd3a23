#!/usr/bin/env python3
"""Compile csrc/mlp_fused.hip to ISA and check the ring loops of the fused ConvNeXt MLP kernels: inside a loop the only vmcnt waits
must be the counted ones written in the source (a compiler-inserted `s_waitcnt vmcnt(0)` between the DMA issue and the MFMAs drains
the LDS ring every stage).  usage: python3 tools/check_mlp_isa.py"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out = os.path.join(tempfile.mkdtemp(), "mlp_fused.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "iseg_amd", "csrc"), "--cuda-device-only", "-S", "-o", out,
           os.path.join(ROOT, "iseg_amd", "csrc", "mlp_fused.hip")]
    subprocess.run(cmd, check=True, capture_output=True)
    text = open(out).read().split("\n")
    bad = 0
    kernel, in_loop, after_issue = None, False, False
    in_asm = False      # between ;;#ASMSTART and ;;#ASMEND: a wait written in the source (the counted ring waits and the tail's vmcnt(0)), not the compiler's
    for ln in text:
        if "#ASMSTART" in ln:
            in_asm = True
        elif "#ASMEND" in ln:
            in_asm = False
        m = re.match(r"^(_ZN\S*convnext_mlp_(fwd|bwd)_kernel\S*):", ln)
        if m:
            kernel, in_loop, after_issue = m.group(1), False, False
            if "ILi384E" in kernel:      # opt-in experiment instances (the backward one spills at 512 registers): not on the product path
                kernel = None
            continue
        if kernel is None:
            continue
        if "s_endpgm" in ln:
            kernel = None
            continue
        if "Loop Header" in ln:
            in_loop, after_issue = True, False
        if in_loop and "global_load_lds" in ln:
            after_issue = True
        if in_loop and after_issue and "v_mfma" in ln:
            after_issue = False
        if in_loop and after_issue and not in_asm and re.search(r"s_waitcnt vmcnt\(0\)", ln):
            print(f"{kernel}: vmcnt(0) between the DMA issue and the first MFMA of a ring stage")
            bad += 1
    print("mlp_fused ISA:", "OK" if not bad else f"{bad} ring drain(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
