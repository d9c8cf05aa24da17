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
