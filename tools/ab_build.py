#!/usr/bin/env python3
"""A/B builds of single translation units: recompile the named csrc units with extra -D defines and relink libiseg_hip.so in place.
usage: python3 tools/ab_build.py "DEFINE1 DEFINE2" unit1 [unit2 ...]      (units without extension, e.g. mlp_fused; "" restores the plain build)"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iseg_amd import build as B  # noqa: E402

defines = [f"-D{d}" for d in sys.argv[1].split() if d]
base = [f for f in B.FLAGS if not f.startswith("-DISEG_ABL") ]
for u in sys.argv[2:]:
    cmd = [B._hipcc()] + base + defines + ["-c", os.path.join(B.CSRC, u + ".hip"), "-o", os.path.join(B.LIBDIR, u + ".o")]
    print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
objs = [os.path.join(B.LIBDIR, s.replace(".hip", ".o")) for s in B.SOURCES]
subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", B.LIB] + objs, check=True)
print("relinked", B.LIB)
