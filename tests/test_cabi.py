"""The C-ABI shared library loads and exports every symbol include/peppa_hip.h declares
(no compute calls: there is no GPU in the CPU test tier), and fails loudly without a device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "peppa_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pf_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = _declared_symbols()
    for must in ("pf_create", "pf_destroy", "pf_load_program", "pf_landmark_forward", "pf_detector_forward",
                 "pf_detect", "pf_landmarks", "pf_run_frames", "pf_last_error"):
        assert must in syms


def test_hip_library_exports_every_declared_symbol(hip_library):
    lib = ctypes.CDLL(hip_library)
    for s in _declared_symbols():
        assert hasattr(lib, s), s
    lib.pf_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.pf_version()


def test_emulator_library_exports_the_same_abi(emu_library):
    lib = ctypes.CDLL(emu_library)
    for s in _declared_symbols():
        assert hasattr(lib, s), s


def test_no_cpu_fallback_without_gpu(hip_library):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from peppa_pig_face_landmark_amd._native import Engine, PeppaHipError
    with pytest.raises(PeppaHipError):
        Engine(0, hip_library)


def test_missing_library_fails_loudly(tmp_path):
    from peppa_pig_face_landmark_amd._native import PeppaHipError, load_library
    with pytest.raises(PeppaHipError):
        load_library(str(tmp_path / "libpeppa_hip.so"))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "peppa_pig_face_landmark_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".inl")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)
                assert "simt_emu" not in src, os.path.join(dirpath, f)


def test_production_library_has_no_ablation_switch():
    """Round-2 verdict: one environment variable (PEPPA_DBG) used to make every conv compute garbage and switch the range guard
    off in the SHIPPING library.  The ablation masks are compiled in only with -DPF_ABLATE=1 (libpeppa_hip_ablate.so, a tool
    build); the production library must not even contain the variable's name."""
    from peppa_pig_face_landmark_amd import build
    lib = build.build_hip()
    with open(lib, "rb") as f:
        blob = f.read()
    assert b"PEPPA_DBG" not in blob
    assert b"PEPPA_RCCL_LIBRARY" in blob          # sanity: environment names the library does read are visible this way


def test_hand_counted_vmcnt_kernels_use_no_scratch(hip_library, tmp_path):
    """The kernels that order their LDS-DMA rings with hand-counted ``s_waitcnt vmcnt(N)`` (k_hero.h, k_sepup.h, k_chain.h, k_hrb.h,
    the unrolled pointwise GEMM) assume that NOTHING but their own requests sits in the vector-memory queue: a register spill puts a
    scratch store / reload there and shifts every count by one (round 5: the mixed-precision split form spilled ONE register of the
    hero kernel).  The code object's metadata must show a zero private segment for every instance of them."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("llvm-objdump / llvm-readelf not found")
    lib = tmp_path / "lib.so"                         # --offloading extracts next to its input
    shutil.copy(hip_library, lib)
    subprocess.run([objdump, "--offloading", str(lib)], check=True, capture_output=True)
    objs = [p for p in tmp_path.iterdir() if "gfx950" in p.name]
    assert objs, "no gfx950 code object in the library"
    seen, bad = 0, []
    for co in objs:
        notes = subprocess.run([readelf, "--notes", str(co)], check=True, capture_output=True, text=True).stdout
        name = None
        for line in notes.splitlines():
            m = re.match(r"\s*\.name:\s+(\S+)", line)
            if m:
                name = m.group(1)
            m = re.match(r"\s*\.private_segment_fixed_size:\s+(\d+)", line)
            if m and name:
                counted = (name.startswith("_Z19conv3x3_hero_kernelILi4ELb1E") or "sepup_pipe_kernel" in name or "basic_chain_kernel" in name
                           or "hr_bottleneck_kernel" in name or re.search(r"conv_gemm_split_kernelILi\d+ELi\d+ELi\d+ELi\d+ELi1ELi0ELin?\d+ELi1ELi0ELi[1-9]", name))
                if counted:
                    seen += 1
                    if int(m.group(1)) != 0:
                        bad.append((name, int(m.group(1))))
    assert seen >= 8, "kernel names changed? only %d hand-counted kernels recognised" % seen
    assert not bad, "hand-counted vmcnt kernels with scratch: %s" % bad


def test_library_has_no_mixed_precision_fma(hip_library, tmp_path):
    """Round 5: with v_fma_mix* selected by the compiler the detector kernels lost the low halves of their splits on MI355X
    (6e-4 of the tensor range against the oracle instead of 5e-5; csrc/pf_intrinsics.h pf_split_lo).  build.py switches the instruction
    family off; this is the check that the flag reached every translation unit."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not found")
    lib = tmp_path / "lib.so"
    shutil.copy(hip_library, lib)
    subprocess.run([objdump, "--offloading", str(lib)], check=True, capture_output=True)
    objs = [p for p in tmp_path.iterdir() if "gfx950" in p.name]
    assert objs
    n_mfma, symbols = 0, ""
    for co in objs:
        asm = subprocess.run([objdump, "-d", str(co)], check=True, capture_output=True, text=True).stdout
        assert "s_endpgm" in asm, "%s did not disassemble as gfx950 code" % co.name
        n_mfma += asm.count("v_mfma_f32_16x16x32")
        symbols += asm
        assert "v_fma_mix" not in asm, "%s contains mixed-precision fma instructions" % co.name
    # the check above is vacuous on an empty or mis-extracted disassembly: the product's matrix instruction and two kernels every
    # build contains must have been seen
    assert n_mfma > 100, "only %d v_mfma_f32_16x16x32 in the disassembly: extraction broken?" % n_mfma
    assert "conv3x3_hero_kernel" in symbols and "mbx_kernel" in symbols, "known kernel symbols missing from the disassembly"
