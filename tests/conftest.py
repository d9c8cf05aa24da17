import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_visible() -> bool:
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_visible():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def student_weights():
    from oracle import synth_weights as sw
    return sw.student_weights()


@pytest.fixture(scope="session")
def detector_weights():
    from oracle import synth_weights as sw
    return sw.detector_weights()


@pytest.fixture(scope="session")
def emu_library():
    """CPU SIMT-emulator build of the engine sources (test infrastructure, see tests/simt_emu)."""
    from tests.simt_emu import build_emu
    if not build_emu.available():
        pytest.skip("host clang not available for the SIMT emulator")
    return build_emu.build_emu()


@pytest.fixture()
def emu_engine(emu_library):
    from peppa_pig_face_landmark_amd._native import Engine
    eng = Engine(0, emu_library)
    yield eng
    eng.close()


@pytest.fixture(scope="session")
def hip_library():
    from peppa_pig_face_landmark_amd import build
    return build.build_hip()


@pytest.fixture()
def gpu_engine(hip_library):
    from peppa_pig_face_landmark_amd._native import Engine
    eng = Engine(0, hip_library)
    yield eng
    eng.close()
