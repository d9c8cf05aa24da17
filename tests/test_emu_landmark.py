"""Kernel-logic parity of the landmark regressor on the CPU SIMT emulator (no GPU needed).

The emulator executes the *same* HIP sources (implicit-GEMM MFMA tiling, fused epilogues, SE /
SCSE / ASPP fusions, fused heat-map arg-max) with gfx950 fragment layouts; the checker is the
oracle.  This guards the indexing and fusion algebra; the real-hardware parity lives in
test_gpu_landmark.py.
"""
import numpy as np
import pytest

from oracle import synth_weights as sw
from peppa_pig_face_landmark_amd.graph.student import build_student_program
from tests import helpers


@pytest.mark.parametrize("size,dtype", [(64, "f32"), (128, "f32"), (64, "f32s"), (128, "f32s")])
def test_student_f32_layers_and_landmarks(emu_engine, student_weights, size, dtype):
    """f32 = exact v_mfma_f32 path; f32s = f32 tensors with split-precision (3 x f16 MFMA) convs."""
    B = 2
    blob, info = build_student_program(student_weights, size, dtype, keep_all=True, debug_full_hm=True)
    emu_engine.load_program(0, blob, B)
    crops = sw.smooth_blob_images(B, size, seed=1000 + size)
    loc, score = emu_engine.landmark_forward(crops)
    oloc, oscore, taps = helpers.oracle_student(student_weights, crops)
    for name in info["tensors"]:
        if name not in taps:
            continue
        ref = helpers.tap_nhwc(taps, name)
        got = helpers.read_engine_tensor(emu_engine, 0, info, name, B, ref.shape[1:], 4)
        rel = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12)
        assert rel < 2e-4, (name, rel)
    # tolerance of the north star is 1e-3 (normalised crop units); the f32 path is ~1e-5
    margins = helpers.heat_margins(taps)
    safe = margins > 1e-3
    d = np.abs(loc - oloc).reshape(B, 98, 2).max(2)
    assert d[safe].max() < 1e-4
    assert np.abs(score - oscore)[safe].max() < 2e-3


def test_student_f32_input_kinds_agree(emu_engine, student_weights):
    """uint8 NHWC input (engine-native) and float32 NCHW /255 input (the ONNX seam) agree."""
    B, size = 1, 64
    blob, _ = build_student_program(student_weights, size, "f32")
    emu_engine.load_program(0, blob, B)
    crops = sw.smooth_blob_images(B, size, seed=5)
    loc8, score8 = emu_engine.landmark_forward(crops)
    xf = (crops.astype(np.float32) / np.float32(255.0)).transpose(0, 3, 1, 2)
    locf, scoref = emu_engine.landmark_forward(np.ascontiguousarray(xf))
    assert np.abs(loc8 - locf).max() < 2e-4
    assert np.abs(score8 - scoref).max() < 2e-3


def test_student_buffer_reuse_matches_keep_all(emu_engine, student_weights):
    """The lifetime-based arena planner must not change results."""
    B, size = 2, 64
    crops = sw.smooth_blob_images(B, size, seed=6)
    outs = []
    for keep in (True, False):
        blob, _ = build_student_program(student_weights, size, "f32", keep_all=keep)
        emu_engine.load_program(0, blob, B)
        outs.append(emu_engine.landmark_forward(crops))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])


def test_student_f16_tracks_oracle(emu_engine, student_weights):
    """f16 storage / f32 accumulate: every feature map within 5% of its range, landmarks equal
    wherever the oracle's arg-max margin exceeds the accumulated f16 error."""
    B, size = 2, 64
    blob, info = build_student_program(student_weights, size, "f16", keep_all=True, debug_full_hm=True)
    emu_engine.load_program(0, blob, B)
    crops = sw.smooth_blob_images(B, size, seed=7)
    loc, score = emu_engine.landmark_forward(crops)
    oloc, oscore, taps = helpers.oracle_student(student_weights, crops)
    ref = helpers.tap_nhwc(taps, "hm")
    got = helpers.read_engine_tensor(emu_engine, 0, info, "hm", B, ref.shape[1:], 8)
    hm_err = np.abs(got - ref).max()
    assert hm_err / np.abs(ref).max() < 0.08
    margins = helpers.heat_margins(taps)
    safe = margins > 4 * hm_err
    if safe.any():
        idx_ok = np.abs(loc - oloc).reshape(B, 98, 2).max(2)[safe]
        assert idx_ok.max() < (4 * hm_err) / 64 + 1e-3


@pytest.mark.parametrize("size", [128, 256])
def test_production_f32s_program_with_fused_decoder_front_end(emu_library, student_weights, size):
    """The production f32s program (arena reuse, fused MBConv/EXPDW blocks, fused 98-channel head and the fused
    DecoderBlock front end: bilinear x2 + concat + depthwise + pointwise in ONE launch) against the oracle, at 16/32-wide
    (size 128) and 32/64-wide (size 256) decoder maps -- the keep_all debug programs of the other tests never fuse it."""
    from peppa_pig_face_landmark_amd._native import Engine
    eng = Engine(0, emu_library)
    try:
        B = 1 if size == 256 else 2
        blob, _ = build_student_program(student_weights, size, "f32s")
        eng.load_program(0, blob, B)
        crops = sw.smooth_blob_images(B, size, seed=1700 + size)
        loc, score = eng.landmark_forward(crops)
        oloc, oscore, taps = helpers.oracle_student(student_weights, crops)
        safe = helpers.heat_margins(taps) > 1e-3
        d = np.abs(loc - oloc).reshape(B, 98, 2).max(2)
        assert safe.mean() > 0.9 and d[safe].max() < 2.5e-4, d[safe].max()     # north-star bound 1e-3; f32 and f32s both sit at ~1e-4 on the emulator (its MFMA sums in another order than torch)
        assert np.abs(score - oscore)[safe].max() < 4e-3          # raw heat-map maxima of O(30): 1e-4 of their range
    finally:
        eng.close()


@pytest.mark.parametrize("batch", [3, 9])
def test_pipelined_decoder_front_end_ragged_batches(emu_library, student_weights, batch):
    """sepup_pipe_kernel (csrc/k_sepup.h) walks its tiles per XCD: workgroup b serves faces b & 7, (b & 7) + 8, ...  With 3
    faces five of the eight lists are empty (those workgroups leave before the first barrier); with 9 faces list 0 holds two
    faces, so its workgroups run their rings across a tile AND a face boundary.  Size 128: 32 x 32 / 128-channel and
    16 x 16 / 256-channel variants, both against the oracle."""
    from peppa_pig_face_landmark_amd._native import Engine
    eng = Engine(0, emu_library)
    try:
        blob, _ = build_student_program(student_weights, 128, "f32s")
        eng.load_program(0, blob, batch)
        crops = sw.smooth_blob_images(batch, 128, seed=1900 + batch)
        loc, score = eng.landmark_forward(crops)
        oloc, oscore, taps = helpers.oracle_student(student_weights, crops)
        safe = helpers.heat_margins(taps) > 1e-3
        d = np.abs(loc - oloc).reshape(batch, 98, 2).max(2)
        assert safe.mean() > 0.9 and d[safe].max() < 2.5e-4, d[safe].max()     # north-star bound 1e-3; f32 and f32s both sit at ~1e-4 on the emulator (its MFMA sums in another order than torch)
        assert np.abs(score - oscore)[safe].max() < 4e-3          # raw heat-map maxima of O(30): 1e-4 of their range
    finally:
        eng.close()


def test_fc_pair_launch_equals_two_fc_launches(emu_engine, student_weights):
    """Round 6: the SE / cSE / ASPP-pool FC pairs run as ONE fc2_kernel launch each (ir.fc_pair, csrc/k_layers.h): same outputs as the
    two-launch form up to the order of the partial sums (different k slicing)."""
    from peppa_pig_face_landmark_amd.graph.student import build_student_program
    crops = sw.smooth_blob_images(3, 128, seed=31)          # 3 faces: a ragged last workgroup (4 faces per workgroup)
    outs, nops = [], []
    for fuse in (True, False):
        blob, info = build_student_program(student_weights, 128, "f32s", fuse_fc_pairs=fuse)
        emu_engine.load_program(0, blob, 3)
        outs.append(emu_engine.landmark_forward(crops))
        nops.append(info["n_ops"])
    # six pairs with small matrices (ir.fc_pair: three stage-2 SE blocks, blocks.4.0's, cSE, ASPP pool) + the cSE squeeze pass, whose sums
    # the decoder front end leaves behind per tile (csrc/k_sepup.h gap_part)
    assert nops[1] - nops[0] == 7
    assert np.abs(outs[0][0] - outs[1][0]).max() < 2e-6 and np.abs(outs[0][1] - outs[1][1]).max() < 2e-4      # (scores are logits of range ~ 20)


@pytest.mark.parametrize("size", [64, 128])
def test_front2_launch_equals_stem_plus_block0(emu_engine, student_weights, size):
    """Round 6: conv_stem + blocks.0.0 as ONE launch (csrc/k_front2.h; the 16-channel stem map stays in LDS) against the two launches it
    replaces, on uint8 crops and through the float NCHW seam of pf_landmark_forward, ragged tiles included (a 64 x 64 crop has a 32 x 32
    stem map: one 32-wide tile column, four 8-row tile rows)."""
    from peppa_pig_face_landmark_amd.graph.student import build_student_program
    crops = sw.smooth_blob_images(2, size, seed=41)
    xf = np.ascontiguousarray((crops.astype(np.float32) / np.float32(255.0)).transpose(0, 3, 1, 2))
    outs = []
    for fuse in (True, False):
        blob, info = build_student_program(student_weights, size, "f32s", fuse_front2=fuse)
        emu_engine.load_program(0, blob, 2)
        outs.append((emu_engine.landmark_forward(crops), emu_engine.landmark_forward(xf)))
    for k in range(2):          # uint8 input, float input
        assert np.abs(outs[0][k][0] - outs[1][k][0]).max() < 1e-5, k
        assert np.abs(outs[0][k][1] - outs[1][k][1]).max() < 1e-3, k


def test_one_product_mix_stays_within_its_budget_emu(emu_engine, student_weights):
    """Round 6, opt-in: decoder.upsampler2.conv2 and the score head on ONE f16 product (csrc/k_hero.h, k_pwhead.h ONEPROD; the emulator
    implements the MFMA's f16 operands and f32 accumulation).  One 256 x 256 face: within 2.5e-4 of the oracle on the margin-safe
    landmarks, the default program within 1e-5, and the two differ (the switch reaches the kernels)."""
    from peppa_pig_face_landmark_amd.graph.student import build_student_program
    crops = sw.smooth_blob_images(1, 256, seed=5)
    oloc, oscore, taps = helpers.oracle_student(student_weights, crops)
    safe = helpers.heat_margins(taps) > 2e-3
    out = {}
    for mix in ((), ("hero", "head")):
        blob, _ = build_student_program(student_weights, 256, "f32s", one_product=mix)
        emu_engine.load_program(0, blob, 1)
        out[mix] = emu_engine.landmark_forward(crops)
    d3 = np.abs(out[()][0] - oloc).reshape(1, 98, 2).max(2)[safe].max()
    d1 = np.abs(out[("hero", "head")][0] - oloc).reshape(1, 98, 2).max(2)[safe].max()
    assert d3 < 1e-5 and d3 < d1 < 2.5e-4, (d3, d1)
