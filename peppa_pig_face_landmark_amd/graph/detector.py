"""yolov5n-0.5 face detector -> packed HIP program.

The reference executes a pre-exported third-party blob (``pretrained/yolov5n-0.5.onnx``,
Skps/config/Skps.yml:4; deepcam-cn/yolov5-face, README.md:24-26) whose contract is pinned at
Skps/core/api/face_detector.py:29-37: ``[1,3,384,640]`` RGB/255 in, ``(15120,16)`` decoded rows
out.  The graph here follows the published ``yolov5n-0.5.yaml`` (StemBlock, ShuffleV2 backbone,
PAN head with C3 blocks, Detect with 3 anchors x 16 outputs per level) with ``Conv`` =
Conv2d + BatchNorm(eps 1e-3) + SiLU folded into fused implicit-GEMM launches.

``weights``: ``{name: ndarray}`` with upstream state_dict names (``model.0.stem_1.conv.weight`` ...).

Layout tricks (all zero-cost views into concat buffers):
  * torch.cat is never materialised by a copy of both halves: producers write channel slices;
  * ShuffleV2's channel_shuffle(groups=2) is folded into the stores: branch outputs are written
    with channel stride 2 (even channels = pass-through / branch1, odd = branch2).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from . import ir

BN_EPS = 1e-3
ANCHORS = [[4, 5, 8, 10, 13, 16], [23, 29, 43, 55, 73, 105], [146, 217, 231, 300, 335, 433]]
STRIDES = [8, 16, 32]
NO = 16
_BACKBONE = [(1, 16, 64, 3), (3, 64, 128, 7), (5, 128, 256, 3)]


def _bn(w, prefix):
    return {k: w[f"{prefix}.{k}"] for k in ("weight", "bias", "running_mean", "running_var")}


def build_detector_program(weights: Dict[str, np.ndarray], input_hw: Tuple[int, int] = (384, 640),
                           dtype: str = "f32", keep_all: bool = False, fuse_units: bool = True, wg_units: bool = True):
    """``wg_units`` (f32s programs): every ShuffleV2Block is ONE workgroup-level launch (csrc/k_det.h); ``fuse_units`` alone
    keeps round 1's wave-per-patch kernel for the stride-1 blocks and runs the stride-2 ones layer by layer."""
    w = weights
    H, W = input_hw
    assert H % 32 == 0 and W % 32 == 0
    pb = ir.ProgramBuilder(dtype, H, W, keep_all=keep_all)

    def fconv(prefix):
        return ir.fold_bn(w[f"{prefix}.conv.weight"], None, _bn(w, f"{prefix}.bn"), BN_EPS)

    def conv(x, prefix, k=1, s=1, out=None, name=""):
        wt, b = fconv(prefix)
        return pb.conv(x, wt, b, "silu", stride=s, pad=k // 2, out=out, out_name=name or prefix)

    # ---- StemBlock ------------------------------------------------------------------------------
    if wg_units and fuse_units and pb.split:
        x = pb.det_stem(*fconv("model.0.stem_1"), *fconv("model.0.stem_2a"), *fconv("model.0.stem_2b"), *fconv("model.0.stem_3"),
                        out_name="model.0")
    else:
        x = None
    wt, b = fconv("model.0.stem_1")
    s1 = pb.stem(wt, b, "silu", out_name="model.0.stem_1") if x is None else -1
    h2, w2 = H // 4, W // 4
    if x is None:
        cat = pb.buffer(h2 * w2 * 32, ir.ELEM_ACT, "stem.cat")
        s2a = conv(s1, "model.0.stem_2a")
        conv(s2a, "model.0.stem_2b", 3, 2, out=pb.view(cat, h2, w2, 16, 0, 32))
        pb.maxpool(s1, out=pb.view(cat, h2, w2, 16, 16, 32))
        x = conv(pb.view(cat, h2, w2, 32, 0, 32), "model.0.stem_3", name="model.0")

    # ---- ShuffleV2 backbone -----------------------------------------------------------------------
    def shuffle_block(x, prefix, inp, oup, stride, name="", into=None):
        """``into = (buffer, channel offset, pixel stride)``: write the block's output straight into a channel slice of a
        concat buffer (the PAN head's torch.cat inputs) instead of a buffer of its own."""
        tx = pb.tensors[x]
        bf = oup // 2
        oh, ow = tx.H // stride, tx.W // stride
        if into is None:
            out_buf, ocoff, old = pb.buffer(oh * ow * oup, ir.ELEM_ACT, name or prefix), 0, oup
        else:
            out_buf, ocoff, old = into
        if wg_units and fuse_units and pb.det_unit_supported(bf, tx.C, stride):
            fb = lambda cw, bn: ir.fold_bn(w[f"{prefix}.{cw}.weight"], None, _bn(w, f"{prefix}.{bn}"), BN_EPS)
            out_view = pb.view(out_buf, oh, ow, oup, ocoff, old, name=name)
            w1, b1 = fb("branch2.0", "branch2.1")
            wd, bd = fb("branch2.3", "branch2.4")
            w2, b2 = fb("branch2.5", "branch2.6")
            if stride == 1:
                pb.det_unit(x, out_view, 1, w1, b1, wd, bd, w2, b2)
            else:
                wd1, bd1 = fb("branch1.0", "branch1.1")
                w3, b3 = fb("branch1.2", "branch1.3")
                pb.det_unit(x, out_view, 2, w1, b1, wd, bd, w2, b2, wd1, bd1, w3, b3)
            return out_view
        even = pb.strided_view(out_buf, oh, ow, bf, ocoff + 0, old)
        odd = pb.strided_view(out_buf, oh, ow, bf, ocoff + 1, old)
        if stride == 1 and fuse_units and pb.shuffle_unit_supported(bf):
            x1 = pb.view(tx.buf, tx.H, tx.W, bf, tx.coff, tx.ld)
            x2 = pb.view(tx.buf, tx.H, tx.W, bf, tx.coff + bf, tx.ld)
            w1, b1 = ir.fold_bn(w[f"{prefix}.branch2.0.weight"], None, _bn(w, f"{prefix}.branch2.1"), BN_EPS)
            wd, bd = ir.fold_bn(w[f"{prefix}.branch2.3.weight"], None, _bn(w, f"{prefix}.branch2.4"), BN_EPS)
            w2, b2 = ir.fold_bn(w[f"{prefix}.branch2.5.weight"], None, _bn(w, f"{prefix}.branch2.6"), BN_EPS)
            pb.shuffle_unit(x2, w1, b1, wd, bd, w2, b2, "silu", odd, 2, x1, even)
            return pb.view(out_buf, oh, ow, oup, ocoff, old, name=name)
        if stride == 1:
            pb.copy(pb.view(tx.buf, tx.H, tx.W, bf, tx.coff, tx.ld), even, out_cs=2)
            x2 = pb.view(tx.buf, tx.H, tx.W, bf, tx.coff + bf, tx.ld)
        else:
            wt, b = ir.fold_bn(w[f"{prefix}.branch1.0.weight"], None, _bn(w, f"{prefix}.branch1.1"), BN_EPS)
            y = pb.dw(x, wt, b, "none", stride=2, pad=1)
            wt, b = ir.fold_bn(w[f"{prefix}.branch1.2.weight"], None, _bn(w, f"{prefix}.branch1.3"), BN_EPS)
            pb.conv(y, wt, b, "silu", out=even, out_cs=2)
            x2 = x
        wt, b = ir.fold_bn(w[f"{prefix}.branch2.0.weight"], None, _bn(w, f"{prefix}.branch2.1"), BN_EPS)
        y = pb.conv(x2, wt, b, "silu")
        wt, b = ir.fold_bn(w[f"{prefix}.branch2.3.weight"], None, _bn(w, f"{prefix}.branch2.4"), BN_EPS)
        y = pb.dw(y, wt, b, "none", stride=stride, pad=1)
        wt, b = ir.fold_bn(w[f"{prefix}.branch2.5.weight"], None, _bn(w, f"{prefix}.branch2.6"), BN_EPS)
        pb.conv(y, wt, b, "silu", out=odd, out_cs=2)
        return pb.view(out_buf, oh, ow, oup, ocoff, old, name=name)

    # ---- concat buffers of the PAN head, allocated up front: every torch.cat input that is produced at the concat's own
    # resolution is WRITTEN into its channel slice by its producer (6 of the 8 copy launches of the straightforward graph are
    # gone); only the two nearest-x2 upsampled inputs are still copied
    h8, w8, h16, w16, h32, w32 = H // 8, W // 8, H // 16, W // 16, H // 32, W // 32
    wg_c3 = wg_units and fuse_units and pb.det_c3_supported(192, "conv") and pb.det_c3_supported(128, "detect")
    if not wg_c3:
        cat9 = pb.buffer(h16 * w16 * 192, ir.ELEM_ACT, "cat9")        # [up(model.7) 64 | model.4 128] @ /16
        cat13 = pb.buffer(h8 * w8 * 128, ir.ELEM_ACT, "cat13")        # [up(model.11) 64 | model.2 64] @ /8
    cat16 = pb.buffer(h16 * w16 * 128, ir.ELEM_ACT, "cat16")      # [model.15 64 | model.11 64] @ /16
    cat19 = pb.buffer(h32 * w32 * 128, ir.ELEM_ACT, "cat19")      # [model.18 64 | model.7 64] @ /32
    # (with the workgroup-level C3 kernels the Upsample + Concat in front of model.10 / model.14 is never materialised: the
    # kernel reads its two sources where they lie)
    into = {} if wg_c3 else {2: (cat13, 64, 128), 4: (cat9, 64, 192)}

    feats = {}
    for li, cin, cout, reps in _BACKBONE:
        x = shuffle_block(x, f"model.{li}", cin, cout, 2, name=f"model.{li}")
        for r in range(reps):
            last = r == reps - 1
            x = shuffle_block(x, f"model.{li + 1}.{r}", cout, cout, 1, name=f"model.{li + 1}" if last else "",
                              into=into.get(li + 1) if last else None)
        feats[li + 1] = x

    # ---- PAN head -------------------------------------------------------------------------------------------------
    def c3(x, prefix, name):
        tx = pb.tensors[x]
        catb = pb.buffer(tx.H * tx.W * 64, ir.ELEM_ACT, prefix + ".cat")
        y1 = conv(x, f"{prefix}.cv1")
        y1 = conv(y1, f"{prefix}.m.0.cv1")
        conv(y1, f"{prefix}.m.0.cv2", 3, 1, out=pb.view(catb, tx.H, tx.W, 32, 0, 64))
        conv(x, f"{prefix}.cv2", out=pb.view(catb, tx.H, tx.W, 32, 32, 64))
        return conv(pb.view(catb, tx.H, tx.W, 64, 0, 64), f"{prefix}.cv3", name=name)

    if wg_c3:
        levels = [(h8, w8), (h16, w16), (h32, w32)]
        nrows = sum(3 * hh * ww for hh, ww in levels)
        rows = pb.buffer(nrows * NO, ir.ELEM_F32, "rows", pinned=True)
        row_starts = [0, 3 * h8 * w8, 3 * h8 * w8 + 3 * h16 * w16]

        def c3_wg(src_a, src_b, up_a, prefix, **kw):
            ws = []
            for part in ("cv1", "cv2", "m.0.cv1", "m.0.cv2", "cv3"):
                ws.extend(fconv(f"{prefix}.{part}"))
            pb.det_c3(src_a, src_b, up_a, *ws, **kw)

        def detect_tail(i, hh, ww):
            raw = pb.tensor(hh, ww, 3 * NO, name=f"model.21.m.{i}") if keep_all else -1
            return dict(tail="detect", w_tail=w[f"model.21.m.{i}.weight"].astype(np.float64), b_tail=w[f"model.21.m.{i}.bias"].astype(np.float64),
                        out2=raw, rows_buf=rows, row0=row_starts[i], det_stride=STRIDES[i], anchors=ANCHORS[i], nrows_total=nrows)

        l7 = conv(feats[6], "model.7", out=pb.view(cat19, h32, w32, 64, 64, 128), name="model.7")
        l11 = pb.view(cat16, h16, w16, 64, 64, 128, name="model.11")
        w11, b11 = fconv("model.11")
        c3_wg(l7, feats[4], True, "model.10", tail="conv", w_tail=w11, b_tail=b11, out2=l11)
        l14 = pb.tensor(h8, w8, 64, name="model.14")
        c3_wg(l11, feats[2], True, "model.14", out=l14, **detect_tail(0, h8, w8))
        conv(l14, "model.15", 3, 2, out=pb.view(cat16, h16, w16, 64, 0, 128), name="model.15")
        l17 = pb.tensor(h16, w16, 64, name="model.17")
        c3_wg(pb.view(cat16, h16, w16, 128, 0, 128), -1, False, "model.17", out=l17, **detect_tail(1, h16, w16))
        conv(l17, "model.18", 3, 2, out=pb.view(cat19, h32, w32, 64, 0, 128), name="model.18")
        l20 = pb.tensor(h32, w32, 64, name="model.20")
        c3_wg(pb.view(cat19, h32, w32, 128, 0, 128), -1, False, "model.20", out=l20, **detect_tail(2, h32, w32))
        blob = pb.finish([rows])
        info = {"tensors": dict(pb.tensor_names), "rows": nrows, "n_ops": len(pb.ops), "dtype": dtype}
        return blob, info

    l7 = conv(feats[6], "model.7", out=pb.view(cat19, h32, w32, 64, 64, 128), name="model.7")
    pb.copy(l7, pb.view(cat9, h16, w16, 64, 0, 192), out_cs=1, up=2)
    l10 = c3(pb.view(cat9, h16, w16, 192, 0, 192), "model.10", "model.10")
    l11 = conv(l10, "model.11", out=pb.view(cat16, h16, w16, 64, 64, 128), name="model.11")
    pb.copy(l11, pb.view(cat13, h8, w8, 64, 0, 128), out_cs=1, up=2)
    l14 = c3(pb.view(cat13, h8, w8, 128, 0, 128), "model.14", "model.14")
    conv(l14, "model.15", 3, 2, out=pb.view(cat16, h16, w16, 64, 0, 128), name="model.15")
    l17 = c3(pb.view(cat16, h16, w16, 128, 0, 128), "model.17", "model.17")
    conv(l17, "model.18", 3, 2, out=pb.view(cat19, h32, w32, 64, 0, 128), name="model.18")
    l20 = c3(pb.view(cat19, h32, w32, 128, 0, 128), "model.20", "model.20")

    # ---- Detect + decode ----------------------------------------------------------------------------------
    levels = [l14, l17, l20]
    nrows = sum(3 * pb.tensors[t].H * pb.tensors[t].W for t in levels)
    rows = pb.buffer(nrows * NO, ir.ELEM_F32, "rows", pinned=True)
    row0 = 0
    for i, t in enumerate(levels):
        o = pb.conv(t, w[f"model.21.m.{i}.weight"].astype(np.float64), w[f"model.21.m.{i}.bias"].astype(np.float64),
                    "none", out_name=f"model.21.m.{i}")
        pb.detdec(o, rows, row0, STRIDES[i], np.asarray(ANCHORS[i], np.float64), nrows)
        row0 += 3 * pb.tensors[t].H * pb.tensors[t].W
    blob = pb.finish([rows])
    info = {"tensors": dict(pb.tensor_names), "rows": nrows, "n_ops": len(pb.ops), "dtype": dtype}
    return blob, info


def detector_param_shapes() -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, kind) of every tensor the detector graph needs, in EXECUTION order (the order the convolutions
    run in ``Model.forward`` of yolov5-face, hence the order of the Conv nodes of an ONNX export)."""
    out: List[Tuple[str, Tuple[int, ...], str]] = []

    def cv(p, cin, cout, k):
        out.extend([(f"{p}.conv.weight", (cout, cin, k, k), "conv"), (f"{p}.bn", (cout,), "bn")])

    def sh(p, inp, oup, stride):
        bf = oup // 2
        if stride > 1:
            out.extend([(f"{p}.branch1.0.weight", (inp, 1, 3, 3), "conv"), (f"{p}.branch1.1", (inp,), "bn"),
                        (f"{p}.branch1.2.weight", (bf, inp, 1, 1), "conv"), (f"{p}.branch1.3", (bf,), "bn")])
        c2 = inp if stride > 1 else bf
        out.extend([(f"{p}.branch2.0.weight", (bf, c2, 1, 1), "conv"), (f"{p}.branch2.1", (bf,), "bn"),
                    (f"{p}.branch2.3.weight", (bf, 1, 3, 3), "conv"), (f"{p}.branch2.4", (bf,), "bn"),
                    (f"{p}.branch2.5.weight", (bf, bf, 1, 1), "conv"), (f"{p}.branch2.6", (bf,), "bn")])

    def c3(p, c1):   # execution order of C3.forward: cv3(cat(m(cv1(x)), cv2(x))) -- the ONNX importer relies on it
        cv(f"{p}.cv1", c1, 32, 1); cv(f"{p}.m.0.cv1", 32, 32, 1); cv(f"{p}.m.0.cv2", 32, 32, 3)
        cv(f"{p}.cv2", c1, 32, 1); cv(f"{p}.cv3", 64, 64, 1)

    cv("model.0.stem_1", 3, 16, 3); cv("model.0.stem_2a", 16, 8, 1); cv("model.0.stem_2b", 8, 16, 3); cv("model.0.stem_3", 32, 16, 1)
    for li, cin, cout, reps in _BACKBONE:
        sh(f"model.{li}", cin, cout, 2)
        for r in range(reps):
            sh(f"model.{li + 1}.{r}", cout, cout, 1)
    cv("model.7", 256, 64, 1); c3("model.10", 192); cv("model.11", 64, 64, 1); c3("model.14", 128)
    cv("model.15", 64, 64, 3); c3("model.17", 128); cv("model.18", 64, 64, 3); c3("model.20", 128)
    for i in range(3):
        out.extend([(f"model.21.m.{i}.weight", (3 * NO, 64, 1, 1), "conv"), (f"model.21.m.{i}.bias", (3 * NO,), "bias")])
    return out


def random_detector_weights(seed: int = 1) -> Dict[str, np.ndarray]:
    """Random-init weights of the exact architecture (benchmarks only; see graph/random_init.py)."""
    rng = np.random.default_rng(seed)
    w: Dict[str, np.ndarray] = {}
    last_var = 1.0
    for name, shape, kind in detector_param_shapes():
        if kind == "conv":
            cout, cin_g, kh, kw = shape
            std = np.sqrt(2.0 / (cout * kh * kw))
            w[name] = (rng.standard_normal(shape) * std).astype(np.float32)
            last_var = max(cin_g * kh * kw * std * std * 0.4, 1e-3)
        elif kind == "bias":
            w[name] = (rng.standard_normal(shape) * 0.05).astype(np.float32)
        else:
            w[f"{name}.weight"] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
            w[f"{name}.bias"] = (rng.standard_normal(shape) * 0.3).astype(np.float32)
            w[f"{name}.running_mean"] = np.zeros(shape, np.float32)
            w[f"{name}.running_var"] = np.full(shape, last_var, np.float32)
    return w
