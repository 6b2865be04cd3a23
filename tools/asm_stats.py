"""Instruction mix / register use of the kernels in one .hip source (device-only -S compile): python tools/asm_stats.py dwconv.hip [filter]"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    path = src if os.path.exists(src) else os.path.join(ROOT, "iseg_amd", "csrc", src)
    out = "/tmp/asm_stats.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{ROOT}/iseg_amd/csrc",
                           "-S", "--cuda-device-only"] + [f"-D{d}" for d in os.environ.get("ASM_DEFINES", "").split()] + ["-o", out, path], stderr=subprocess.DEVNULL)
    text = open(out).read()
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)^\s*\.amdhsa_kernel \1\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M):
        name, body, meta = m.groups()
        if flt not in name:
            continue
        ops = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        c = collections.Counter(ops)
        filt = subprocess.run(["c++filt", name], capture_output=True, text=True)
        demangled = filt.stdout.strip() or name
        print(demangled[:150])
        print("  instructions", len(ops), " valu", sum(v for k, v in c.items() if k.startswith("v_")), " lds", sum(v for k, v in c.items() if k.startswith("ds_")),
              " waitcnt", c.get("s_waitcnt", 0))
        print("  top:", ", ".join(f"{k}={v}" for k, v in c.most_common(14)))
        for key in ("next_free_vgpr", "accum_offset", "private_segment_fixed_size", "group_segment_fixed_size"):
            mm = re.search(rf"\.amdhsa_{key} (\S+)", meta)
            if mm:
                print(f"  {key} = {mm.group(1)}")
        sp = re.search(rf"; ScratchSize: (\d+)", text[m.end():m.end() + 3000])
        occ = re.search(rf"; Occupancy: (\d+)", text[m.end():m.end() + 3000])
        print("  scratch", sp.group(1) if sp else "?", " occupancy", occ.group(1) if occ else "?")


if __name__ == "__main__":
    main()
