"""Student landmark regressor -> packed HIP program.

Graph source (reference checkout): ``TRAIN/face_landmark/lib/core/base_trainer/model.py``
``Net`` :247-298 (timm ``mobilenetv3_large_100`` features, out_indices [0,1,2,4], output_stride 16,
``blocks[6] = Identity``), ``Decoder`` :212-244, ``ASPP`` :64-96, ``DecoderBlock`` :133-196,
``SCSEModule`` :117-130, ``hm`` head :271 and the in-graph decode ``COTRAIN.postp`` :511-554 --
i.e. exactly what ``tools/convert_to_onnx.py:28,54-61`` exports as ``kps_student.onnx``.

``weights`` is a flat ``{name: ndarray}`` with the reference's state_dict names relative to
``COTRAIN.student`` (``encoder.conv_stem.weight``, ``decoder.aspp.conv1.weight``, ``hm.bias`` ...),
so a real checkpoint (``torch.load(...)`` -> numpy) drops in unchanged.

Fusions performed here (all exact re-associations of the eval-mode graph):
  * every BatchNorm folded into the conv in front of it;
  * ASPP: the post-concat BN+ReLU is per-channel, so it is folded into each branch conv; the
    pooled branch (GAP -> 1x1 -> BN -> ReLU -> broadcast) is constant over the image, so its
    contribution to the 1x1 ``project`` conv is a per-face bias vector computed by two tiny FCs;
  * SE / SCSE gates are computed on pooled vectors and applied inside the consuming kernels;
  * heat-map head: only the 98 score channels go through the GEMM, with a fused running arg-max;
    the 2x98 offset channels are evaluated at the arg-max pixel only (same arithmetic, 2/3 of the
    head's MACs never executed).  ``debug_full_hm=True`` builds the unfused 294-channel head too.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from . import ir

NUM_POINTS = 98

# timm arch_def of mobilenetv3_large_100 (kind, kernel, stride, expand, out_ch, se, act)
_STAGES = [
    [("ds", 3, 1, 1.0, 16, False, "relu")],
    [("ir", 3, 2, 4.0, 24, False, "relu"), ("ir", 3, 1, 3.0, 24, False, "relu")],
    [("ir", 5, 2, 3.0, 40, True, "relu"), ("ir", 5, 1, 3.0, 40, True, "relu"), ("ir", 5, 1, 3.0, 40, True, "relu")],
    [("ir", 3, 2, 6.0, 80, False, "hswish"), ("ir", 3, 1, 2.5, 80, False, "hswish"),
     ("ir", 3, 1, 2.3, 80, False, "hswish"), ("ir", 3, 1, 2.3, 80, False, "hswish")],
    [("ir", 3, 1, 6.0, 112, True, "hswish"), ("ir", 3, 1, 6.0, 112, True, "hswish")],
    [("ir", 5, 2, 6.0, 160, True, "hswish"), ("ir", 5, 1, 6.0, 160, True, "hswish"),
     ("ir", 5, 1, 6.0, 160, True, "hswish")],
]


def _bn(w: Dict[str, np.ndarray], prefix: str) -> Dict[str, np.ndarray]:
    return {k: w[f"{prefix}.{k}"] for k in ("weight", "bias", "running_mean", "running_var")}


def build_decoder_and_head(pb: "ir.ProgramBuilder", w: Dict[str, np.ndarray], encx4: int, encx8: int, encx16: int,
                           input_size: int, keep_all: bool, debug_full_hm: bool, one_product=()):
    """Decoder (ASPP + two DecoderBlocks, model.py:212-244) + hm head + fused decode; shared by the
    Student (mobilenetv3 features 24/40/160 ch) and the Teacher (hrnet_w18 features 128/256/512 ch)."""
    h16 = input_size // 16
    c16 = pb.tensors[encx16].C

    # ---- ASPP (model.py:64-96) ----------------------------------------------------------------
    a = "decoder.aspp"
    s_cat, t_cat = ir.bn_affine(_bn(w, f"{a}.bn_act.0"))        # BN over the 256-channel concat
    cat_buf = pb.buffer(h16 * h16 * 192, ir.ELEM_ACT, "aspp.cat")
    for j, (name, pad, dil) in enumerate((("conv1", 0, 1), ("conv2", 2, 2), ("conv3", 4, 4))):
        wj = w[f"{a}.{name}.weight"].astype(np.float64) * s_cat[64 * j:64 * j + 64].reshape(-1, 1, 1, 1)
        view = pb.view(cat_buf, h16, h16, 64, 64 * j, 192, name=f"{a}.{name}")
        pb.conv(encx16, wj, t_cat[64 * j:64 * j + 64], "relu", pad=pad, dil=dil, out=view)
    cat = pb.view(cat_buf, h16, h16, 192, 0, 192, name=f"{a}.cat192")
    pooled = pb.gap(encx16)
    wp, bp = ir.fold_bn(w[f"{a}.fm_pool.pool.1.weight"], None, _bn(w, f"{a}.fm_pool.pool.2"))
    wproj, bproj = ir.fold_bn(w[f"{a}.project.0.weight"], None, _bn(w, f"{a}.project.1"))
    fbias = pb.fc_pair(pooled, wp.reshape(64, -1), bp, "relu", wproj[:, 192:256, 0, 0], None, "none",
                       scale2=s_cat[192:256], shift2=t_cat[192:256], act1b="relu")
    x16 = pb.conv(cat, wproj[:, :192], bproj, "relu", fbias_buf=fbias, out_name=f"{a}.out")

    # ---- decoder blocks (model.py:133-196) -------------------------------------------------------
    def decoder_block(lo, skip, name, second, att):
        wd, bd = ir.fold_bn(w[f"{name}.conv1.0.conv_dw.0.weight"], w[f"{name}.conv1.0.conv_dw.0.bias"],
                            _bn(w, f"{name}.conv1.0.conv_dw.1"))
        wp, bp = ir.fold_bn(w[f"{name}.conv1.0.conv_pw.weight"], None, _bn(w, f"{name}.conv1.1"))
        parts = -1
        if pb.split and not keep_all:
            # one launch: upsample + concat + depthwise + pointwise (the 280/296-channel tensors never reach HBM).  When the block ends in
            # the SCSE attention, the same launch leaves the per-tile channel sums of its output behind: the squeeze pass (a 268 MB read
            # per 256 faces) disappears and the cSE FC pair adds the partial sums up (round 6)
            if att and not second:
                w1 = w[f"{name}.attention2.cSE.1.weight"]
                if pb.sepconv_up_can_sum(lo, skip, wp.shape[0]) and pb.fc_pair_fuses(wp.shape[0], w1.shape[0], wp.shape[0]):
                    x, parts = pb.sepconv_up(lo, skip, wd, bd, wp, bp, "relu", out_name=f"{name}.pw", gap_parts=True)
            if parts < 0:
                x = pb.sepconv_up(lo, skip, wd, bd, wp, bp, "relu", out_name=f"{name}.pw")
        else:
            x = pb.upcat(lo, skip, out_name=f"{name}.cat")
            x = pb.dw(x, wd, bd, "none", pad=1, out_name=f"{name}.dw")
            x = pb.conv(x, wp, bp, "relu", out_name=f"{name}.pw")
        if second:
            wt, b = ir.fold_bn(w[f"{name}.conv2.0.weight"], w[f"{name}.conv2.0.bias"], _bn(w, f"{name}.conv2.1"))
            x = pb.conv(x, wt, b, "relu", pad=1, out_name=f"{name}.conv2", products=1 if "hero" in one_product else 3)
        if att:
            w1, w2 = w[f"{name}.attention2.cSE.1.weight"], w[f"{name}.attention2.cSE.3.weight"]
            tx = pb.tensors[x]
            pooled, kw = (parts, {"nparts": tx.H * tx.W // 128, "xscale": 1.0 / (tx.H * tx.W)}) if parts >= 0 else (pb.gap(x), {})
            cse = pb.fc_pair(pooled, w1.reshape(w1.shape[0], -1), w[f"{name}.attention2.cSE.1.bias"], "relu",
                             w2.reshape(w2.shape[0], -1), w[f"{name}.attention2.cSE.3.bias"], "sigmoid", **kw)
            x = pb.scse(x, cse, w[f"{name}.attention2.sSE.0.weight"], float(w[f"{name}.attention2.sSE.0.bias"][0]),
                        out_name=f"{name}.scse")
        return x

    decx8 = decoder_block(x16, encx8, "decoder.upsampler1", False, True)
    decx4 = decoder_block(decx8, encx4, "decoder.upsampler2", True, False)

    # ---- heat-map head + decode (model.py:271,295,511-554) ---------------------------------------
    hw = w["hm.weight"].astype(np.float64)
    hb = w["hm.bias"].astype(np.float64)
    h4 = input_size // 4
    info = {"hm_full": -1}
    if debug_full_hm:
        info["hm_full"] = pb.conv(decx4, hw, hb, "none", out_name="hm")
    bm, _, warps_m = ir.CONV_CFGS[0]
    if pb.conv_uses_split(128):
        warps_m *= 2       # the split-precision kernels run 8 waves (2x the M-waves) per workgroup
    assert (h4 * h4) % bm == 0, "heat-map area must be a multiple of the GEMM pixel tile"
    nslots = (h4 * h4 // bm) * warps_m
    val = pb.buffer(NUM_POINTS * nslots, ir.ELEM_F32, "amax_val")
    idx = pb.buffer(NUM_POINTS * nslots, ir.ELEM_I32, "amax_idx")
    dummy = pb.tensor(h4, h4, ir._round_up(NUM_POINTS, pb.ve), buf=pb.buffer(pb.ve, ir.ELEM_ACT, "hm.unused"),
                      coff=0, ld=ir._round_up(NUM_POINTS, pb.ve))
    pb.conv(decx4, hw[:NUM_POINTS], hb[:NUM_POINTS], "none", out=dummy, amax=(val, idx, NUM_POINTS),
            store_out=False, cfg=0, products=1 if "head" in one_product else 3)
    loc, score = pb.hmdec(val, idx, decx4, hw[NUM_POINTS:, :, 0, 0], hb[NUM_POINTS:], NUM_POINTS, nslots)
    return loc, score, info


def build_student_program(weights: Dict[str, np.ndarray], input_size: int = 256, dtype: str = "f16",
                          keep_all: bool = False, debug_full_hm: bool = False, fuse_mbconv: bool = True,
                          fuse_mbx: Optional[bool] = None, mbx_se: Optional[str] = None, mbx_waves: int = 16, fuse_fc_pairs: bool = True,
                          fuse_front2: bool = True, one_product=()):
    """Returns (blob: bytes, info: dict).  ``info['tensors']`` maps layer names to tensor ids for
    ``pf_read_tensor`` (only meaningful with ``keep_all=True``)."""
    assert input_size % 64 == 0, "input size must be a multiple of 64 (heat-map tile = 128 pixels)"
    if fuse_mbx is None:       # the input-stationary block kernels of stages 3-5 (16 x 16 maps at 256 x 256; csrc/k_mbx.h): on with the other fusions
        fuse_mbx = fuse_mbconv     # mbx_se: None = ir.mbx's choice ("store"), or "recompute" / "store" for every SE block (A/B aid)
    w = weights
    pb = ir.ProgramBuilder(dtype, input_size, input_size, keep_all=keep_all)
    pb.fuse_fc_pairs = bool(fuse_fc_pairs)      # SE / cSE / ASPP-pool FC pairs as ONE launch each (ir.fc_pair); False: the round-5 two-launch form (A/B aid)

    # ---- encoder (timm MobileNetV3Features; output_stride 16 => stage 5 runs dilated) ---------
    wt, b = ir.fold_bn(w["encoder.conv_stem.weight"], None, _bn(w, "encoder.bn1"))
    # fuse_front2 (f32s programs, default): conv_stem + blocks.0.0 in ONE shallow launch (csrc/k_front2.h, two barriers): 0.174 against
    # 0.094 + 0.160 ms per 256 crops (profiles/r06_run11_ub_front2.txt)
    fuse_front2 = fuse_mbconv and fuse_front2 and not keep_all and pb.front2_supported()
    if fuse_front2:
        f = lambda cw, bn: ir.fold_bn(w[f"encoder.blocks.{cw}.weight"], None, _bn(w, f"encoder.blocks.{bn}"))
        x = pb.front2(wt, b, "hswish", *f("0.0.conv_dw", "0.0.bn1"), *f("0.0.conv_pw", "0.0.bn2"), out_name="encoder.blocks.0.0.out")
    else:
        x = pb.stem(wt, b, "hswish", out_name="encoder.stem")
    cin, cur_stride, cur_dil = 16, 2, 1
    feats = {}
    for si, stack in enumerate(_STAGES):
        for bi, (kind, k, s, e, cout, se, act) in enumerate(stack):
            if fuse_front2 and (si, bi) == (0, 0):                # inside front2
                continue
            if bi >= 1:
                s = 1
            next_dil = cur_dil
            if s > 1:
                if cur_stride * s > 16:
                    next_dil, s = cur_dil * s, 1
                else:
                    cur_stride *= s
            pad = ((s - 1) + cur_dil * (k - 1)) // 2
            p = f"encoder.blocks.{si}.{bi}"
            inp = x
            skip = (s == 1 and cin == cout)
            if kind == "ds" and fuse_mbconv and pb.dsconv_supported(cin, k, s, cur_dil, cout):
                wd, bd = ir.fold_bn(w[f"{p}.conv_dw.weight"], None, _bn(w, f"{p}.bn1"))
                wp, bp = ir.fold_bn(w[f"{p}.conv_pw.weight"], None, _bn(w, f"{p}.bn2"))
                x = pb.dsconv(x, wd, bd, wp, bp, act, res=inp if skip else -1, out_name=f"{p}.out")
            elif kind == "ds":
                wt, b = ir.fold_bn(w[f"{p}.conv_dw.weight"], None, _bn(w, f"{p}.bn1"))
                x = pb.dw(x, wt, b, act, stride=s, pad=pad, dil=cur_dil, out_name=f"{p}.dw")
                wt, b = ir.fold_bn(w[f"{p}.conv_pw.weight"], None, _bn(w, f"{p}.bn2"))
                x = pb.conv(x, wt, b, "none", res=inp if skip else -1, out_name=f"{p}.out")
            elif fuse_mbx and kind == "ir" and pb.mbx_supported(x, k, s, pad, cur_dil, cout, se):
                # stages 3-5 at 16 x 16: the whole block with the face's input stationary in registers (csrc/k_mbx.h).  An SE block is,
                # by default (ir.mbx se_mode "store"), a squeeze pass that also STORES the activated depthwise map -> the SE FC pair ->
                # the layer-wise gated projection reading that map back; mbx_se="recompute" is the A/B alternative (second pass
                # recomputes expand + depthwise, the expanded tensor never in HBM -- measured slower for every block)
                we, be = ir.fold_bn(w[f"{p}.conv_pw.weight"], None, _bn(w, f"{p}.bn1"))
                wd, bd = ir.fold_bn(w[f"{p}.conv_dw.weight"], None, _bn(w, f"{p}.bn2"))
                wl, bl = ir.fold_bn(w[f"{p}.conv_pwl.weight"], None, _bn(w, f"{p}.bn3"))
                fcs = None
                if se:
                    rd, ex = w[f"{p}.se.conv_reduce.weight"], w[f"{p}.se.conv_expand.weight"]
                    fcs = (rd.reshape(rd.shape[0], rd.shape[1]), w[f"{p}.se.conv_reduce.bias"],
                           ex.reshape(ex.shape[0], ex.shape[1]), w[f"{p}.se.conv_expand.bias"])
                x = pb.mbx(x, we, be, wd, bd, wl, bl, act, pad=pad, dil=cur_dil, res=inp if skip else -1, se_fcs=fcs,
                           se_mode=mbx_se, waves=mbx_waves, out_name=f"{p}.out", dw_name=f"{p}.dw")
            elif (fuse_mbconv and not se and pb.mbconv_supported(cin, k, s, cur_dil, cout)
                  and not pb.expdw_supported(pb.tensors[x].H, pb.tensors[x].W, k, s, pad, cur_dil)):
                we, be = ir.fold_bn(w[f"{p}.conv_pw.weight"], None, _bn(w, f"{p}.bn1"))
                wd, bd = ir.fold_bn(w[f"{p}.conv_dw.weight"], None, _bn(w, f"{p}.bn2"))
                wl, bl = ir.fold_bn(w[f"{p}.conv_pwl.weight"], None, _bn(w, f"{p}.bn3"))
                x = pb.mbconv(x, we, be, wd, bd, wl, bl, act, stride=s, pad=pad, dil=cur_dil,
                              res=inp if skip else -1, out_name=f"{p}.out")
            else:
                ti = pb.tensors[x]
                pooled = -1
                if fuse_mbconv and pb.expdw_supported(ti.H, ti.W, k, s, pad, cur_dil, cin):
                    we, be = ir.fold_bn(w[f"{p}.conv_pw.weight"], None, _bn(w, f"{p}.bn1"))
                    wd, bd = ir.fold_bn(w[f"{p}.conv_dw.weight"], None, _bn(w, f"{p}.bn2"))
                    x, pooled = pb.expdw(x, we, be, wd, bd, act, pad=pad, dil=cur_dil, stride=s, want_gap=bool(se), out_name=f"{p}.dw")
                else:
                    wt, b = ir.fold_bn(w[f"{p}.conv_pw.weight"], None, _bn(w, f"{p}.bn1"))
                    x = pb.conv(x, wt, b, act, out_name=f"{p}.pw")
                    wt, b = ir.fold_bn(w[f"{p}.conv_dw.weight"], None, _bn(w, f"{p}.bn2"))
                    x = pb.dw(x, wt, b, act, stride=s, pad=pad, dil=cur_dil, out_name=f"{p}.dw")
                gate = -1
                if se:
                    if pooled < 0:
                        pooled = pb.gap(x)
                    rd = w[f"{p}.se.conv_reduce.weight"]
                    ex = w[f"{p}.se.conv_expand.weight"]
                    gate = pb.fc_pair(pooled, rd.reshape(rd.shape[0], rd.shape[1]), w[f"{p}.se.conv_reduce.bias"], "relu",
                                      ex.reshape(ex.shape[0], ex.shape[1]), w[f"{p}.se.conv_expand.bias"], "hsigmoid")
                wt, b = ir.fold_bn(w[f"{p}.conv_pwl.weight"], None, _bn(w, f"{p}.bn3"))
                x = pb.conv(x, wt, b, "none", res=inp if skip else -1, gate_buf=gate, out_name=f"{p}.out")
            cur_dil = next_dil
            cin = cout
        feats[si] = x
    # one_product (opt-in, f32s programs; NOT the parity-grade default): names of layers to run on ONE f16 product instead of the split's
    # three -- "hero" = decoder.upsampler2.conv2 (42 % of the dense MACs; alone 4.9e-5 of the oracle's landmarks on the synthetic weights,
    # profiles/r06_student_precision_study.txt), "head" = the 98 score channels of the heat-map conv (7.4e-5 alone; the offsets at the
    # arg-max stay exact f32 in hm_decode)
    loc, score, info = build_decoder_and_head(pb, w, feats[1], feats[2], feats[5], input_size, keep_all, debug_full_hm, tuple(one_product))
    blob = pb.finish([loc, score])
    info.update({"tensors": dict(pb.tensor_names), "input_size": input_size, "dtype": dtype,
                 "n_ops": len(pb.ops), "const_bytes": len(pb.consts)})
    return blob, info
