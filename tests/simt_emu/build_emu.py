"""Build the CPU SIMT-emulator flavour of the engine (TEST INFRASTRUCTURE ONLY).

Same sources as libpeppa_hip.so, compiled by the host clang against tests/simt_emu/include
(a fibre-based stand-in for the HIP runtime + gfx950 MFMA semantics).  Used by the ``not gpu``
tests to check kernel indexing / fusion logic where no GPU exists; never loaded by the product.
"""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
CSRC = os.path.join(ROOT, "peppa_pig_face_landmark_amd", "csrc")
OUT_DIR = os.path.join(ROOT, "tests", "_build")
OUT = os.path.join(OUT_DIR, "libpeppa_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def available() -> bool:
    return os.path.exists(CLANG)


def build_emu(force: bool = False, flavour: str = "") -> str:
    """``flavour="asan"``: the same sources under AddressSanitizer (tests/test_jpeg_decode.py runs the host-side JPEG parser of
    that build on hostile files in a subprocess with the sanitizer runtime preloaded)."""
    os.makedirs(OUT_DIR, exist_ok=True)
    out = OUT if not flavour else os.path.join(OUT_DIR, "libpeppa_emu_%s.so" % flavour)
    extra = {"": [], "asan": ["-fsanitize=address", "-fno-omit-frame-pointer", "-shared-libasan", "-O1"]}[flavour]
    srcs = [os.path.join(CSRC, "engine.cpp"), os.path.join(CSRC, "mbx_launch.cpp"), os.path.join(HERE, "emu_runtime.cpp")]
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [
        os.path.join(HERE, "emu_runtime.cpp"), os.path.join(HERE, "include", "pf_intrinsics.h"),
        os.path.join(HERE, "include", "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "peppa_hip.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    cmd = [CLANG, "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-pthread",
           "-I", os.path.join(HERE, "include"), "-I", CSRC] + os.environ.get("PEPPA_EMU_CFLAGS", "").split() + extra + srcs + ["-o", out]
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build_emu(force=True))
