"""Build ``libpeppa_hip.so`` (HIP, gfx950) in-tree.

    python -m peppa_pig_face_landmark_amd.build [--force]

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container; the
resulting .so is git-ignored but travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libpeppa_hip.so")
SOURCES = ["engine.cpp"]
DEPS = ["engine.cpp", "pipeline.inl", "k_conv_gemm.h", "k_layers.h", "k_prepost.h", "pf_common.h",
        "pf_intrinsics.h", "pf_program.h", os.path.join("..", "..", "include", "peppa_hip.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build_hip(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
           "-I", CSRC] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        print("[peppa-hip] " + " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv))
