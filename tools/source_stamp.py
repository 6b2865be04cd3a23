#!/usr/bin/env python3
"""Content stamp of everything that decides what the flagship step launches: the HIP sources and the host code between bench.py and the
C ABI.  profiles/rNN_pmc.json carries the stamp of the tree its counters were taken on; bench.py recomputes it and reports the PMC-derived
fractions only when the two agree (a later kernel change must not silently report stale traffic).  `python tools/source_stamp.py` prints it."""
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["include/iseg_hip.h", "iseg_amd/functional.py", "iseg_amd/kernels.py", "iseg_amd/trainer.py", "iseg_amd/heads.py", "iseg_amd/nn.py",
         "iseg_amd/param_store.py", "iseg_amd/backbones/convnext.py", "iseg_amd/layers/aspp.py", "iseg_amd/layers/model_builder.py",
         "iseg_amd/layers/core_model_ext.py", "iseg_amd/optimizers/modern.py"]


def source_stamp():
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "iseg_amd", "csrc")
    paths = [os.path.join("iseg_amd", "csrc", f) for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h"))] + FILES
    for rel in paths:
        h.update(rel.encode())
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def git_head():
    """the commit the tree sits on, when there is a repository (the GPU box receives a snapshot without .git)"""
    try:
        r = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10)
        return r.stdout.strip() or None if r.returncode == 0 else None
    except (OSError, subprocess.SubprocessError):
        return None


if __name__ == "__main__":
    print(source_stamp(), git_head())
