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
SOURCES = ["engine.cpp", "mbx_launch.cpp"]
# No SLP vectoriser anywhere: on gfx950 a v_pk_fma_f32 costs what two v_fma_f32 cost, so pairing scalar f32 work buys nothing
# and the v_mov shuffles that feed the pairs are pure loss (the VALU-heavy depthwise / unit kernels measure 5-13 % faster
# without it, the whole pipeline ~2 %: profiles/r05_run19_noslp_whole_library.txt, r05_run20_noslp_headline_ab.txt).  For
# k_mbx.h's depthwise taps it is a requirement, not a preference (see csrc/mbx_launch.cpp).
COMMON_FLAGS = ["-fno-slp-vectorize",
                # no mixed-precision fma instructions (v_fma_mix*): with them selected by the compiler the DETECTOR kernels measured
                # 6e-4 of their range against the oracle instead of 5e-5 (csrc/pf_intrinsics.h pf_split_lo; tools/det_parity.py)
                "-Xclang", "-target-feature", "-Xclang", "-fma-mix-insts"]
SOURCE_FLAGS = {}            # per-source extras
STAMP = os.path.join(HERE, "libpeppa_hip.srchash")


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def source_hash() -> str:
    """Content hash of every file the library is built from (csrc/* and the public header): the staleness test
    must not depend on mtimes, which a repository snapshot copied to another machine does not preserve."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(HERE, "..", "include", "peppa_hip.h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(repr((COMMON_FLAGS, sorted(SOURCE_FLAGS.items()))).encode())
    return h.hexdigest()


def needs_build() -> bool:
    if not os.path.exists(OUT) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


ABLATE_OUT = os.path.join(HERE, "libpeppa_hip_ablate.so")
STRICT_OUT = os.path.join(HERE, "libpeppa_hip_strict.so")
# tool / test flavours of the same sources (never loaded by the product: _native.DEFAULT_LIBRARY is libpeppa_hip.so)
FLAVOURS = {"": (OUT, []), "ablate": (ABLATE_OUT, ["-DPF_ABLATE=1"]), "strict": (STRICT_OUT, ["-DPF_STRICT_WAITS=1"])}


def build_hip(force: bool = False, verbose: bool = True, ablate: bool = False, flavour: str = "") -> str:
    """``ablate=True`` (= ``flavour="ablate"``) builds the TOOL flavour (``libpeppa_hip_ablate.so``, -DPF_ABLATE=1): the same
    sources with the timing ablations of the GEMM kernels compiled in and PEPPA_DBG honoured (tools/ab_env.py; wrong results
    by construction).  ``flavour="strict"`` builds the TEST flavour ``libpeppa_hip_strict.so`` (-DPF_STRICT_WAITS=1): every
    partial ``s_waitcnt vmcnt(N)`` in front of a raw barrier (pf_wait_vm_barrier<N>, the hand-counted LDS-DMA rings) drains to
    vmcnt(0) instead, so a miscounted N shows up as production != strict in tests/test_gpu_race_net.py.  The production
    library has neither switch."""
    flavour = "ablate" if ablate else flavour
    out, defs = FLAVOURS[flavour]
    stamp = STAMP if not flavour else out + ".srchash"
    if not force:
        if flavour and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == source_hash():
            return out
        if not flavour and not needs_build():
            return out
    base = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-I", CSRC] + COMMON_FLAGS + defs
    objs, procs = [], []
    for src in SOURCES:                       # one object per translation unit, compiled side by side, then one link
        obj = out + "." + os.path.splitext(src)[0] + ".o"
        cmd = base + SOURCE_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[peppa-hip] " + " ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out]
    if verbose:
        print("[peppa-hip] " + " ".join(link), flush=True)
    subprocess.run(link, check=True)
    for obj in objs:
        os.remove(obj)
    with open(stamp, "w") as f:
        f.write(source_hash() + "\n")
    return out


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, ablate="--ablate" in sys.argv, flavour="strict" if "--strict" in sys.argv else ""))
