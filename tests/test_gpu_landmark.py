"""Parity of the HIP landmark regressor on a real MI355X (through the C ABI) against the oracle.

Tolerance: the north star asks for landmarks within 1e-3 (normalised crop units, the ONNX
output units of model.py:549-552) on identical 256x256 crops.  The f32 path is checked at 1e-4.
"""
import numpy as np
import pytest

from oracle import synth_weights as sw
from peppa_pig_face_landmark_amd.graph.student import build_student_program
from tests import helpers

pytestmark = pytest.mark.gpu

LANDMARK_TOL = 1e-3  # north-star tolerance; asserted values below are tighter where stated


def _layer_report(eng, info, taps, batch, ve):
    worst = ("", 0.0)
    for name in info["tensors"]:
        if name not in taps:
            continue
        ref = helpers.tap_nhwc(taps, name)
        got = helpers.read_engine_tensor(eng, 0, info, name, batch, ref.shape[1:], ve)
        rel = float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12))
        if rel > worst[1]:
            worst = (name, rel)
    return worst


@pytest.mark.parametrize("size,batch,dtype", [(256, 4, "f32"), (128, 3, "f32"), (256, 4, "f32s"), (128, 3, "f32s")])
def test_student_f32_matches_oracle(gpu_engine, student_weights, size, batch, dtype):
    """f32: exact v_mfma_f32_16x16x4_f32 convs.  f32s: f32 tensors, split-precision convs
    (hi/lo f16 operands, 3 x v_mfma_f32_16x16x32_f16, f32 accumulate) -- same tolerance."""
    blob, info = build_student_program(student_weights, size, dtype, keep_all=True, debug_full_hm=True)
    gpu_engine.load_program(0, blob, batch)
    crops = sw.smooth_blob_images(batch, size, seed=4000 + size)
    loc, score = gpu_engine.landmark_forward(crops)
    oloc, oscore, taps = helpers.oracle_student(student_weights, crops)
    name, rel = _layer_report(gpu_engine, info, taps, batch, 4)
    assert rel < 5e-4, (name, rel)
    margins = helpers.heat_margins(taps)
    safe = margins > 2e-3
    assert safe.mean() > 0.9
    d = np.abs(loc - oloc).reshape(batch, 98, 2).max(2)
    assert d[safe].max() < 1e-4 < LANDMARK_TOL
    assert np.abs(score - oscore)[safe].max() < 5e-3
    # unsafe (near-tie) landmarks may pick the other of two equal-height cells; report, don't hide
    flips = int((d[~safe] > LANDMARK_TOL).sum())
    n_unsafe = int((~safe).sum())
    print(f"near-tie landmarks: {n_unsafe}, of which flipped: {flips}")
    # bounded flip rate: even among the near-ties (oracle top-1/top-2 margin <= 2e-3) a flip needs the
    # engine's heat-map error to exceed half the margin, so at most a small share may move
    assert flips <= max(1, n_unsafe // 4), (flips, n_unsafe)
    assert flips <= 0.005 * safe.size + 1, (flips, safe.size)


def test_student_one_product_hero_mix_stays_within_its_budget(gpu_engine, student_weights):
    """Round 6, OPT-IN (bench.py --mix hero; never the parity-grade default): decoder.upsampler2.conv2 -- 42 % of the Student's dense
    MACs -- on ONE f16 product per 32 k instead of the split's three (csrc/k_hero.h ONEPROD).  The oracle study
    (tools/teacher_precision_study.py --model student, profiles/r06_student_precision_study.txt) puts this layer alone at 4.9e-5 of the
    oracle's landmarks; the engine must stay within 2.5e-4 (north star: 1e-3) on the margin-safe landmarks, flip none of them, and
    differ from the default program only behind that conv."""
    size, batch = 256, 6
    crops = sw.smooth_blob_images(batch, size, seed=4256)
    oloc, oscore, taps = helpers.oracle_student(student_weights, crops)
    margins = helpers.heat_margins(taps)
    safe = margins > 2e-3
    res = {}
    for mix in ((), ("hero",), ("hero", "head")):
        blob, info = build_student_program(student_weights, size, "f32s", one_product=mix)
        gpu_engine.load_program(0, blob, batch)
        res[mix] = gpu_engine.landmark_forward(crops)
    err = {mix: np.abs(r[0] - oloc).reshape(batch, 98, 2).max(2) for mix, r in res.items()}
    print("max |loc - oracle| on the %d margin-safe landmarks of %d: " % (int(safe.sum()), safe.size) +
          "; ".join("%s %.2e" % ("+".join(m) or "three products", e[safe].max()) for m, e in err.items()))
    assert err[()][safe].max() < 1e-4
    for mix in (("hero",), ("hero", "head")):
        assert err[mix][safe].max() < 2.5e-4, mix                      # no margin-safe landmark moved to another cell either (a cell is 1.6e-2)
        assert np.abs(res[mix][1] - oscore)[safe].max() < 3e-2, mix     # scores: heat-map logits of range ~ 20
    assert not np.array_equal(res[()][0], res[("hero",)][0])           # the switches reached the kernels
    assert not np.array_equal(res[("hero",)][1], res[("hero", "head")][1])


def test_student_f32_production_program_equals_debug_program(gpu_engine, student_weights):
    """Arena reuse + fused 98-channel head give the same answers as the keep-all debug build."""
    size, batch = 256, 4
    crops = sw.smooth_blob_images(batch, size, seed=11)
    blob, _ = build_student_program(student_weights, size, "f32", keep_all=True, debug_full_hm=True)
    gpu_engine.load_program(0, blob, batch)
    a = gpu_engine.landmark_forward(crops)
    blob, _ = build_student_program(student_weights, size, "f32")
    gpu_engine.load_program(0, blob, batch)
    b = gpu_engine.landmark_forward(crops)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("dtype", ["f32", "f32s"])
def test_student_f32_batch_256_is_batch_independent(gpu_engine, student_weights, dtype):
    """BASELINE config 2 size (256 crops): every face's result is independent of its batch-mates
    (size-independent property; the oracle cannot run 256 faces in seconds)."""
    size = 256
    blob, _ = build_student_program(student_weights, size, dtype)
    gpu_engine.load_program(0, blob, 256)
    base = sw.smooth_blob_images(8, size, seed=21)
    big = np.concatenate([base] * 32, 0)
    perm = np.random.default_rng(3).permutation(256)
    loc, score = gpu_engine.landmark_forward(big[perm])
    loc8, score8 = gpu_engine.landmark_forward(base)
    assert np.array_equal(loc, loc8[perm % 8])
    assert np.array_equal(score, score8[perm % 8])
    oloc, _, taps = helpers.oracle_student(student_weights, base)
    safe = helpers.heat_margins(taps) > 2e-3
    assert np.abs(loc8 - oloc).reshape(8, 98, 2).max(2)[safe].max() < 1e-4


def test_student_f32_nchw_float_input_seam(gpu_engine, student_weights):
    """ONNXEngine-level seam: float32 NCHW /255 input (face_landmark.py:44-47) == uint8 NHWC path."""
    size, batch = 256, 2
    blob, _ = build_student_program(student_weights, size, "f32")
    gpu_engine.load_program(0, blob, batch)
    crops = sw.smooth_blob_images(batch, size, seed=31)
    loc8, score8 = gpu_engine.landmark_forward(crops)
    xf = np.ascontiguousarray((crops.astype(np.float32) / np.float32(255.0)).transpose(0, 3, 1, 2))
    locf, scoref = gpu_engine.landmark_forward(xf)
    assert np.abs(loc8 - locf).max() < 1e-4
    assert np.abs(score8 - scoref).max() < 5e-3


def test_student_f16_fast_mode(gpu_engine, student_weights):
    """f16 storage / f16 MFMA / f32 accumulate fast mode.  NOT the parity-grade path: on the
    noise-like synthetic weights the f16 rounding noise is amplified by the network (the sSE gate of
    SCSE has |w| ~ 1.4 on 256 channels) so arg-max flips are expected; this test only checks that
    the mode runs, stays finite, tracks the oracle's heat-maps in the large, and that landmarks
    whose oracle arg-max margin exceeds the measured heat-map error agree.  Stats are printed."""
    size, batch = 256, 4
    blob, info = build_student_program(student_weights, size, "f16", keep_all=True, debug_full_hm=True)
    gpu_engine.load_program(0, blob, batch)
    crops = sw.smooth_blob_images(batch, size, seed=41)
    loc, score = gpu_engine.landmark_forward(crops)
    assert np.isfinite(loc).all() and np.isfinite(score).all()
    oloc, oscore, taps = helpers.oracle_student(student_weights, crops)
    ref = helpers.tap_nhwc(taps, "hm")
    got = helpers.read_engine_tensor(gpu_engine, 0, info, "hm", batch, ref.shape[1:], 8)
    hm_err = float(np.abs(got - ref).max())
    corr = float(np.corrcoef(got.ravel(), ref.ravel())[0, 1])
    margins = helpers.heat_margins(taps)
    d = np.abs(loc - oloc).reshape(batch, 98, 2).max(2)
    safe = margins > 4 * hm_err
    print(f"f16: hm max err {hm_err:.3f} (range {np.abs(ref).max():.1f}), corr {corr:.5f}; "
          f"landmarks within 1e-3: {(d < 1e-3).mean():.3f}; margin-safe fraction {safe.mean():.3f}")
    assert corr > 0.98
    if safe.any():
        assert d[safe].max() < 4 * hm_err / 64 + 1e-3


@pytest.mark.parametrize("dtype", ["f32", "f32s"])
def test_teacher_256_matches_oracle(gpu_engine, dtype):
    """BASELINE config 5 model: Teacher@256 (HRNet-W18 encoder, model.py:302-345) vs the oracle."""
    import torch
    from oracle import landmark_net as ln
    from oracle import teacher_net as tn
    from peppa_pig_face_landmark_amd.graph.teacher import build_teacher_program
    weights = sw.teacher_weights()
    size, batch = 256, 3
    blob, info = build_teacher_program(weights, size, dtype, keep_all=True, debug_full_hm=True)
    gpu_engine.load_program(0, blob, batch)
    crops = sw.smooth_blob_images(batch, size, seed=77)
    loc, score = gpu_engine.landmark_forward(crops)
    x = torch.from_numpy(crops.astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
    taps = {}
    with torch.no_grad():
        oloc, oscore = tn.teacher_forward(ln.to_torch(weights), x, taps)
    worst = ("", 0.0)
    for name in info["tensors"]:
        if name in taps and taps[name].dtype.is_floating_point:
            ref = helpers.tap_nhwc(taps, name)
            got = helpers.read_engine_tensor(gpu_engine, 0, info, name, batch, ref.shape[1:], 4)
            rel = float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12))
            worst = max(worst, (name, rel), key=lambda t: t[1])
    assert worst[1] < 1e-3, worst
    safe = helpers.heat_margins(taps) > 2e-3
    d = np.abs(loc - oloc.numpy()).reshape(batch, 98, 2).max(2)
    assert d[safe].max() < 2e-4 < LANDMARK_TOL
    # production build (arena reuse, fused head / decoder front end) gives the same landmarks
    blob, _ = build_teacher_program(weights, size, dtype)
    gpu_engine.load_program(0, blob, batch)
    loc2, _ = gpu_engine.landmark_forward(crops)
    assert np.abs(loc2 - oloc.numpy()).reshape(batch, 98, 2).max(2)[safe].max() < 2e-4


def test_teacher_one_product_hero_head_mix_stays_within_its_budget(gpu_engine):
    """BASELINE config 5 names fp16 MFMA; round 5 asked for the mix of layers that tolerates ONE f16 product within 2.5e-4.  The oracle study
    (profiles/r06_teacher_precision_study.txt) finds four groups -- 21.6 % of the Teacher's dense MACs -- of which the engine has
    one-product kernels for two: the decoder's hero conv and the score head (opt-in: build_teacher_program(one_product=("hero", "head")))."""
    import torch
    from oracle import landmark_net as ln
    from oracle import teacher_net as tn
    from peppa_pig_face_landmark_amd.graph.teacher import build_teacher_program
    weights = sw.teacher_weights()
    size, batch = 256, 4
    crops = sw.smooth_blob_images(batch, size, seed=77)
    x = torch.from_numpy(crops.astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
    taps = {}
    with torch.no_grad():
        oloc, oscore = tn.teacher_forward(ln.to_torch(weights), x, taps)
    safe = helpers.heat_margins(taps) > 2e-3
    errs = {}
    for mix in ((), ("hero", "head")):
        blob, _ = build_teacher_program(weights, size, "f32s", one_product=mix)
        gpu_engine.load_program(0, blob, batch)
        loc, score = gpu_engine.landmark_forward(crops)
        errs[mix] = np.abs(loc - oloc.numpy()).reshape(batch, 98, 2).max(2)[safe].max()
    print("teacher: three products %.2e; hero + head on one product %.2e (%d margin-safe landmarks of %d)" % (errs[()], errs[("hero", "head")], int(safe.sum()), safe.size))
    assert errs[()] < 2e-4
    assert errs[("hero", "head")] < 2.5e-4


def test_teacher_f16_fast_mode_is_measured_not_claimed(gpu_engine):
    """BASELINE config 5 names fp16 MFMA.  f16 storage + f16 MFMA (f32 accumulate) runs the Teacher, but whether it meets the
    1e-3 landmark bound is a property of the weights: this measures it on the synthetic set (heat-map error, fraction of
    landmarks within 1e-3, arg-max flips) and asserts only what the mode promises -- finite outputs that track the oracle."""
    import torch
    from oracle import landmark_net as ln
    from oracle import teacher_net as tn
    from peppa_pig_face_landmark_amd.graph.teacher import build_teacher_program
    weights = sw.teacher_weights()
    size, batch = 256, 3
    blob, info = build_teacher_program(weights, size, "f16", keep_all=True, debug_full_hm=True)
    gpu_engine.load_program(0, blob, batch)
    crops = sw.smooth_blob_images(batch, size, seed=77)
    loc, score = gpu_engine.landmark_forward(crops)
    assert np.isfinite(loc).all() and np.isfinite(score).all()
    x = torch.from_numpy(crops.astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
    taps = {}
    with torch.no_grad():
        oloc, oscore = tn.teacher_forward(ln.to_torch(weights), x, taps)
    ref = helpers.tap_nhwc(taps, "hm")
    got = helpers.read_engine_tensor(gpu_engine, 0, info, "hm", batch, ref.shape[1:], 8)
    hm_err = float(np.abs(got - ref).max())
    corr = float(np.corrcoef(got.ravel(), ref.ravel())[0, 1])
    d = np.abs(loc - oloc.numpy()).reshape(batch, 98, 2).max(2)
    margins = helpers.heat_margins(taps)
    print(f"teacher f16: hm max err {hm_err:.4f} (range {np.abs(ref).max():.1f}), corr {corr:.6f}; landmarks within 1e-3: "
          f"{(d < 1e-3).mean():.3f}, within 1/64 (one heat-map cell): {(d < 1.0 / 64 + 1e-3).mean():.3f}, max {d.max():.4f}; "
          f"arg-max margins above 4x the heat-map error: {(margins > 4 * hm_err).mean():.3f}")
    assert corr > 0.98
    safe = margins > 4 * hm_err
    if safe.any():
        assert d[safe].max() < 4 * hm_err / 64 + 1e-3
