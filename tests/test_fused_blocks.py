"""Fused encoder blocks (csrc/k_mbconv.h, expdw epilogue of csrc/k_conv_gemm.h) against the layer-by-layer
program of the same weights and against the oracle.

The fused program replaces, per inverted-residual block (timm InvertedResidual, Student encoder
model.py:252-264), "pointwise expand -> depthwise -> [SE] -> pointwise project" by
  * one MBCONV launch (blocks without SE, 3x3 depthwise), or
  * one EXPDW launch (expand + depthwise + SE squeeze) followed by the gated projection,
so every block output ("encoder.blocks.i.j.out") must agree with the unfused program to f32 rounding.
Batch sizes that do not fill the last workgroup (3 faces, 4 whole images per workgroup at 8x8) exercise
the ragged-tile guards."""
import numpy as np
import pytest

from oracle import synth_weights as sw
from peppa_pig_face_landmark_amd.graph import ir
from peppa_pig_face_landmark_amd.graph.student import build_student_program
from tests import helpers


def _op_codes(blob: bytes):
    import struct
    hdr = struct.unpack_from("<16i", blob, 0)
    n_bufs, n_tensors, n_ops = hdr[3], hdr[4], hdr[5]
    off = 64 + n_bufs * 16 + n_tensors * 32
    return [struct.unpack_from("<i", blob, off + 160 * i)[0] for i in range(n_ops)]


def _compare_programs(eng, weights, size, batch):
    crops = sw.smooth_blob_images(batch, size, seed=4200 + size + batch)
    outs = {}
    for fuse in (False, True):
        blob, info = build_student_program(weights, size, "f32s", keep_all=True, fuse_mbconv=fuse)
        codes = _op_codes(blob)
        assert (ir.OP_MBCONV in codes) == fuse and (ir.OP_EXPDW in codes) == fuse
        eng.load_program(0, blob, batch)
        loc, score = eng.landmark_forward(crops)
        outs[fuse] = (loc, score)
    return crops, outs


def _check(eng, weights, size, batch):
    crops, outs = _compare_programs(eng, weights, size, batch)
    loc0, score0 = outs[False]
    loc1, score1 = outs[True]
    oloc, oscore, taps = helpers.oracle_student(weights, crops)
    # every block output of the fused program against the oracle's taps (the unfused program passes the same check
    # in test_emu_landmark / test_gpu_landmark)
    blob, info = build_student_program(weights, size, "f32s", keep_all=True, fuse_mbconv=True)
    eng.load_program(0, blob, batch)
    eng.landmark_forward(crops)
    checked = 0
    for name in info["tensors"]:
        if name not in taps or not name.startswith("encoder.blocks."):
            continue
        ref = helpers.tap_nhwc(taps, name)
        got = helpers.read_engine_tensor(eng, 0, info, name, batch, ref.shape[1:], 4)
        rel = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12)
        assert rel < 2e-4, (name, rel)
        checked += 1
    assert checked >= 15
    safe = helpers.heat_margins(taps) > 1e-3
    d = np.abs(loc1 - loc0).reshape(batch, 98, 2).max(2)
    assert d[safe].max() < 1e-4          # fused vs unfused program
    d = np.abs(loc1 - oloc).reshape(batch, 98, 2).max(2)
    assert d[safe].max() < 2e-4          # fused program vs oracle (north-star tolerance is 1e-3)
    assert np.abs(score1 - score0)[safe].max() < 2e-3


@pytest.mark.parametrize("size,batch", [(64, 3), (128, 2), (256, 1)])
def test_fused_blocks_match_unfused_emu(emu_engine, student_weights, size, batch):
    _check(emu_engine, student_weights, size, batch)


@pytest.mark.gpu
@pytest.mark.parametrize("size,batch", [(128, 3), (256, 5)])
def test_fused_blocks_match_unfused_gpu(gpu_engine, student_weights, size, batch):
    _check(gpu_engine, student_weights, size, batch)
