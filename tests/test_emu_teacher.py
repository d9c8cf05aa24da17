"""Teacher (HRNet-W18 encoder) program on the CPU SIMT emulator vs the oracle restatement."""
import numpy as np
import pytest
import torch

from oracle import landmark_net as ln
from oracle import synth_weights as sw
from oracle import teacher_net as tn
from peppa_pig_face_landmark_amd.graph.teacher import build_teacher_program
from tests import helpers


@pytest.fixture(scope="module")
def teacher_weights():
    return sw.teacher_weights()


def oracle_teacher(weights, crops):
    W = ln.to_torch(weights)
    x = torch.from_numpy(crops.astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
    taps = {}
    with torch.no_grad():
        loc, score = tn.teacher_forward(W, x, taps)
    return loc.numpy(), score.numpy(), taps


@pytest.mark.parametrize("dtype", ["f32", "f32s"])
def test_teacher_layers_and_landmarks(emu_engine, teacher_weights, dtype):
    B, size = 2, 64
    blob, info = build_teacher_program(teacher_weights, size, dtype, keep_all=True, debug_full_hm=True)
    emu_engine.load_program(0, blob, B)
    crops = sw.smooth_blob_images(B, size, seed=5)
    loc, score = emu_engine.landmark_forward(crops)
    oloc, oscore, taps = oracle_teacher(teacher_weights, crops)
    for name in info["tensors"]:
        if name in taps and taps[name].dtype.is_floating_point:
            ref = helpers.tap_nhwc(taps, name)
            got = helpers.read_engine_tensor(emu_engine, 0, info, name, B, ref.shape[1:], 4)
            assert np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12) < 3e-4, name
    safe = helpers.heat_margins(taps) > 1e-3
    assert np.abs(loc - oloc).reshape(B, 98, 2).max(2)[safe].max() < 1e-4
    assert np.abs(score - oscore)[safe].max() < 3e-3


def test_teacher_cost_model_matches_reference_readme(teacher_weights):
    """README.md:37: Teacher params 11.53 M(i).  The inference graph built here omits the unused 4th
    'incre' head (1024 ch @ /32, not in out_indices): add its parameters back for the comparison."""
    params = sum(int(np.prod(s)) for n, s, k in tn.param_inventory() if k not in ("bn_mean", "bn_var"))
    incre3 = 144 * 256 + 256 * 256 * 9 + 256 * 1024 + 144 * 1024 + 2 * (256 + 256 + 1024 + 1024)
    assert abs((params + incre3) / 1024 ** 2 - 11.53) < 0.03
