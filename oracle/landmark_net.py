"""Oracle: torch-CPU restatement of the Student landmark regressor.  TEST INFRASTRUCTURE ONLY.

What is restated, and from where (paths relative to /root/reference):

* decoder / heads / heat-map decode: ``TRAIN/face_landmark/lib/core/base_trainer/model.py``
    SeparableConv2d :15-43, ASPPPooling :46-61, ASPP :64-96, SCSEModule :117-130,
    DecoderBlock :133-196, Decoder :212-244, Net :247-298, COTRAIN.postp :511-554,
    inference branch of COTRAIN.forward :562-568.
  This part is PINNED: ``tests/golden/make_golden.py`` runs the reference's own classes
  (``oracle/ref_import.py``) on the same weights and the outputs agree (see
  ``tests/test_oracle_pinned.py``).

* encoder: timm ``mobilenetv3_large_100`` with ``features_only=True, out_indices=[0,1,2,4],
  output_stride=16`` and ``blocks[6] = Identity`` (model.py:252-264).  timm==0.6.11
  (requirements.txt:9) is NOT vendored in the reference, so this is a restatement of the
  published architecture (arch_def of ``_gen_mobilenet_v3`` + ``EfficientNetBuilder``):
  PARITY UNPINNED.  Cross-checks that do hold: MAC and parameter totals reproduce the
  reference README table (README.md:34-37), see ``count_macs_params``.

The network is written functionally over a flat ``dict[str, Tensor]`` whose keys are the
reference's ``state_dict`` names relative to ``COTRAIN.student`` (so weights can be moved
between this oracle and the reference's module tree verbatim).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
NUM_POINTS = 98


# --------------------------------------------------------------------------------------
# architecture description (timm arch_def, decoded by hand)
# --------------------------------------------------------------------------------------
def _make_divisible(v: float, divisor: int = 8, round_limit: float = 0.9) -> int:
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < round_limit * v:
        new_v += divisor
    return new_v


# (kind, kernel, stride, expand, out_ch, se, act) ; stride is the arch_def stride before
# the output_stride=16 rewrite.
_ARCH = [
    [("ds", 3, 1, 1.0, 16, False, "relu")],
    [("ir", 3, 2, 4.0, 24, False, "relu"), ("ir", 3, 1, 3.0, 24, False, "relu")],
    [("ir", 5, 2, 3.0, 40, True, "relu")] + [("ir", 5, 1, 3.0, 40, True, "relu")] * 2,
    [("ir", 3, 2, 6.0, 80, False, "hswish"), ("ir", 3, 1, 2.5, 80, False, "hswish"),
     ("ir", 3, 1, 2.3, 80, False, "hswish"), ("ir", 3, 1, 2.3, 80, False, "hswish")],
    [("ir", 3, 1, 6.0, 112, True, "hswish")] * 2,
    [("ir", 5, 2, 6.0, 160, True, "hswish")] + [("ir", 5, 1, 6.0, 160, True, "hswish")] * 2,
]
STEM_CH = 16
ENCODER_OUT_CHANNELS = [16, 24, 40, 160]  # model.py:264 (without the leading image "3")
FEATURE_AFTER_STAGE = [0, 1, 2, 5]  # timm feature stages 1,2,3,7 with blocks[6]=Identity


class BlockSpec:
    __slots__ = ("name", "kind", "k", "stride", "dil", "pad", "cin", "mid", "cout", "se_rd",
                 "act", "skip")

    def __repr__(self):  # pragma: no cover - debugging aid
        return "BlockSpec(" + ", ".join(f"{s}={getattr(self, s)}" for s in self.__slots__) + ")"


def encoder_blocks(output_stride: int = 16) -> List[List[BlockSpec]]:
    """Decode _ARCH the way timm's EfficientNetBuilder does (stride -> dilation rewrite)."""
    stages = []
    cin = STEM_CH
    cur_stride = 2  # after the stem
    cur_dil = 1
    for si, stack in enumerate(_ARCH):
        blocks = []
        for bi, (kind, k, s, e, cout, se, act) in enumerate(stack):
            if bi >= 1:
                s = 1
            next_dil = cur_dil
            if s > 1:
                if cur_stride * s > output_stride:
                    next_dil = cur_dil * s
                    s = 1
                else:
                    cur_stride *= s
            b = BlockSpec()
            b.name = f"encoder.blocks.{si}.{bi}"
            b.kind, b.k, b.stride, b.dil = kind, k, s, cur_dil
            b.pad = ((s - 1) + cur_dil * (k - 1)) // 2
            b.cin = cin
            b.mid = cin if kind == "ds" else _make_divisible(cin * e)
            b.cout = cout
            b.se_rd = _make_divisible(b.mid * 0.25) if se else 0
            b.act = act
            b.skip = (s == 1 and cin == cout)
            cur_dil = next_dil
            cin = cout
            blocks.append(b)
        stages.append(blocks)
    return stages


# --------------------------------------------------------------------------------------
# parameter inventory: name -> (shape, kind)
# --------------------------------------------------------------------------------------
def _bn_entries(prefix: str, ch: int):
    return [(f"{prefix}.weight", (ch,), "bn_gamma"), (f"{prefix}.bias", (ch,), "bn_beta"),
            (f"{prefix}.running_mean", (ch,), "bn_mean"), (f"{prefix}.running_var", (ch,), "bn_var")]


def param_inventory(include_dead_fc: bool = True) -> List[Tuple[str, Tuple[int, ...], str]]:
    """All tensors of ``COTRAIN.student`` (names relative to it), in definition order."""
    inv: List[Tuple[str, Tuple[int, ...], str]] = []
    inv.append(("encoder.conv_stem.weight", (STEM_CH, 3, 3, 3), "conv"))
    inv += _bn_entries("encoder.bn1", STEM_CH)
    for stage in encoder_blocks():
        for b in stage:
            p = b.name
            if b.kind == "ds":
                inv.append((f"{p}.conv_dw.weight", (b.mid, 1, b.k, b.k), "conv"))
                inv += _bn_entries(f"{p}.bn1", b.mid)
                inv.append((f"{p}.conv_pw.weight", (b.cout, b.mid, 1, 1), "conv"))
                inv += _bn_entries(f"{p}.bn2", b.cout)
            else:
                inv.append((f"{p}.conv_pw.weight", (b.mid, b.cin, 1, 1), "conv"))
                inv += _bn_entries(f"{p}.bn1", b.mid)
                inv.append((f"{p}.conv_dw.weight", (b.mid, 1, b.k, b.k), "conv"))
                inv += _bn_entries(f"{p}.bn2", b.mid)
                if b.se_rd:
                    inv.append((f"{p}.se.conv_reduce.weight", (b.se_rd, b.mid, 1, 1), "conv"))
                    inv.append((f"{p}.se.conv_reduce.bias", (b.se_rd,), "bias"))
                    inv.append((f"{p}.se.conv_expand.weight", (b.mid, b.se_rd, 1, 1), "conv"))
                    inv.append((f"{p}.se.conv_expand.bias", (b.mid,), "bias"))
                inv.append((f"{p}.conv_pwl.weight", (b.cout, b.mid, 1, 1), "conv"))
                inv += _bn_entries(f"{p}.bn3", b.cout)
    c16, c8, c4 = ENCODER_OUT_CHANNELS[3], ENCODER_OUT_CHANNELS[2], ENCODER_OUT_CHANNELS[1]
    a = "decoder.aspp"
    inv.append((f"{a}.conv1.weight", (64, c16, 1, 1), "conv"))
    inv.append((f"{a}.conv2.weight", (64, c16, 3, 3), "conv"))
    inv.append((f"{a}.conv3.weight", (64, c16, 3, 3), "conv"))
    inv += _bn_entries(f"{a}.bn_act.0", 256)
    inv.append((f"{a}.fm_pool.pool.1.weight", (64, c16, 1, 1), "conv"))
    inv += _bn_entries(f"{a}.fm_pool.pool.2", 64)
    inv.append((f"{a}.project.0.weight", (256, 256, 1, 1), "conv"))
    inv += _bn_entries(f"{a}.project.1", 256)
    for name, cin, cout, second, att in (("decoder.upsampler1", 256 + c8, 256, False, True),
                                         ("decoder.upsampler2", 256 + c4, 128, True, False)):
        inv.append((f"{name}.conv1.0.conv_dw.0.weight", (cin, 1, 3, 3), "conv"))
        inv.append((f"{name}.conv1.0.conv_dw.0.bias", (cin,), "bias"))
        inv += _bn_entries(f"{name}.conv1.0.conv_dw.1", cin)
        inv.append((f"{name}.conv1.0.conv_pw.weight", (cout, cin, 1, 1), "conv"))
        inv += _bn_entries(f"{name}.conv1.1", cout)
        if second:
            inv.append((f"{name}.conv2.0.weight", (cout, cout, 3, 3), "conv"))
            inv.append((f"{name}.conv2.0.bias", (cout,), "bias"))
            inv += _bn_entries(f"{name}.conv2.1", cout)
        if att:
            inv.append((f"{name}.attention2.cSE.1.weight", (cout // 4, cout, 1, 1), "conv"))
            inv.append((f"{name}.attention2.cSE.1.bias", (cout // 4,), "bias"))
            inv.append((f"{name}.attention2.cSE.3.weight", (cout, cout // 4, 1, 1), "conv"))
            inv.append((f"{name}.attention2.cSE.3.bias", (cout,), "bias"))
            inv.append((f"{name}.attention2.sSE.0.weight", (1, cout, 1, 1), "conv"))
            inv.append((f"{name}.attention2.sSE.0.bias", (1,), "bias"))
    if include_dead_fc:
        inv.append(("fc.weight", (7, 640), "linear"))
        inv.append(("fc.bias", (7,), "bias"))
    inv.append(("hm.weight", (NUM_POINTS * 3, 128, 1, 1), "conv"))
    inv.append(("hm.bias", (NUM_POINTS * 3,), "bias"))
    return inv


# --------------------------------------------------------------------------------------
# functional forward
# --------------------------------------------------------------------------------------
# When not None, ``_bn`` runs in calibration mode: it measures the per-channel statistics of
# its input, stores them into W as running_mean / running_var, then normalises with them
# (used once by oracle.synth_weights to give the synthetic network trained-like BN statistics).
_CALIBRATING = None


def _bn(W, prefix, x):
    if _CALIBRATING is not None:
        dims = (0, 2, 3)
        mean = x.mean(dims)
        var = x.var(dims, unbiased=False)
        # round through float32 so every machine derives bit-identical statistics
        W[f"{prefix}.running_mean"] = mean.float().to(x.dtype)
        W[f"{prefix}.running_var"] = var.float().clamp_min(1e-3).to(x.dtype)
        _CALIBRATING.append(prefix)
    return F.batch_norm(x, W[f"{prefix}.running_mean"], W[f"{prefix}.running_var"],
                        W[f"{prefix}.weight"], W[f"{prefix}.bias"], False, 0.0, BN_EPS)


def _act(x, kind):
    if kind == "relu":
        return F.relu(x)
    if kind == "hswish":
        return x * F.relu6(x + 3.0) / 6.0
    if kind == "none":
        return x
    raise ValueError(kind)


def _hsigmoid(x):
    return F.relu6(x + 3.0) / 6.0


def _tap(taps, name, t):
    if taps is not None:
        taps[name] = t


def encoder_forward(W: Dict[str, torch.Tensor], x: torch.Tensor, taps=None) -> List[torch.Tensor]:
    """timm MobileNetV3Features.forward (bottleneck feature location).  UNPINNED restatement."""
    x = F.conv2d(x, W["encoder.conv_stem.weight"], None, stride=2, padding=1)
    x = _act(_bn(W, "encoder.bn1", x), "hswish")
    _tap(taps, "encoder.stem", x)
    feats = []
    for si, stage in enumerate(encoder_blocks()):
        for b in stage:
            p = b.name
            inp = x
            if b.kind == "ds":
                x = F.conv2d(x, W[f"{p}.conv_dw.weight"], None, b.stride, b.pad, b.dil, groups=b.mid)
                x = _act(_bn(W, f"{p}.bn1", x), b.act)
                _tap(taps, f"{p}.dw", x)
                x = _bn(W, f"{p}.bn2", F.conv2d(x, W[f"{p}.conv_pw.weight"]))
            else:
                x = _act(_bn(W, f"{p}.bn1", F.conv2d(x, W[f"{p}.conv_pw.weight"])), b.act)
                _tap(taps, f"{p}.pw", x)
                x = F.conv2d(x, W[f"{p}.conv_dw.weight"], None, b.stride, b.pad, b.dil, groups=b.mid)
                x = _act(_bn(W, f"{p}.bn2", x), b.act)
                _tap(taps, f"{p}.dw", x)
                if b.se_rd:
                    s = x.mean((2, 3), keepdim=True)
                    s = F.relu(F.conv2d(s, W[f"{p}.se.conv_reduce.weight"], W[f"{p}.se.conv_reduce.bias"]))
                    g = _hsigmoid(F.conv2d(s, W[f"{p}.se.conv_expand.weight"], W[f"{p}.se.conv_expand.bias"]))
                    _tap(taps, f"{p}.se_gate", g)
                    x = x * g
                x = _bn(W, f"{p}.bn3", F.conv2d(x, W[f"{p}.conv_pwl.weight"]))
            if b.skip:
                x = x + inp
            _tap(taps, f"{p}.out", x)
        if si in FEATURE_AFTER_STAGE:
            feats.append(x)
    return feats


def decoder_forward(W, feats: List[torch.Tensor], taps=None) -> torch.Tensor:
    """Decoder.forward (model.py:232-244) -> decx4 (128 @ S/4)."""
    encx2, encx4, encx8, encx16 = feats
    a = "decoder.aspp"
    # ASPP (model.py:85-96); the three atrous convs carry no BN of their own (:70-73)
    f1 = F.conv2d(encx16, W[f"{a}.conv1.weight"])
    f2 = F.conv2d(encx16, W[f"{a}.conv2.weight"], None, 1, 2, 2)
    f4 = F.conv2d(encx16, W[f"{a}.conv3.weight"], None, 1, 4, 4)
    # ASPPPooling (model.py:46-61): GAP -> 1x1 -> BN -> ReLU -> nearest broadcast
    g = encx16.mean((2, 3), keepdim=True)
    g = F.relu(_bn(W, f"{a}.fm_pool.pool.2", F.conv2d(g, W[f"{a}.fm_pool.pool.1.weight"])))
    g = g.expand(-1, -1, encx16.shape[2], encx16.shape[3])
    cat = torch.cat([f1, f2, f4, g], 1)
    cat = F.relu(_bn(W, f"{a}.bn_act.0", cat))
    _tap(taps, "decoder.aspp.cat", cat)
    x16 = F.relu(_bn(W, f"{a}.project.1", F.conv2d(cat, W[f"{a}.project.0.weight"])))
    _tap(taps, "decoder.aspp.out", x16)

    def block(x, skip, name, second, att):
        # DecoderBlock.forward (model.py:181-196); F.interpolate default align_corners=False
        x = F.interpolate(x, scale_factor=2, mode="bilinear")
        x = torch.cat([x, skip], 1)
        _tap(taps, f"{name}.cat", x)
        c = x.shape[1]
        x = F.conv2d(x, W[f"{name}.conv1.0.conv_dw.0.weight"], W[f"{name}.conv1.0.conv_dw.0.bias"],
                     1, 1, 1, groups=c)
        x = _bn(W, f"{name}.conv1.0.conv_dw.1", x)
        _tap(taps, f"{name}.dw", x)
        x = F.conv2d(x, W[f"{name}.conv1.0.conv_pw.weight"])
        x = F.relu(_bn(W, f"{name}.conv1.1", x))
        _tap(taps, f"{name}.pw", x)
        if second:
            x = F.conv2d(x, W[f"{name}.conv2.0.weight"], W[f"{name}.conv2.0.bias"], 1, 1)
            x = F.relu(_bn(W, f"{name}.conv2.1", x))
            _tap(taps, f"{name}.conv2", x)
        if att:  # SCSEModule (model.py:117-130)
            s = x.mean((2, 3), keepdim=True)
            s = F.relu(F.conv2d(s, W[f"{name}.attention2.cSE.1.weight"], W[f"{name}.attention2.cSE.1.bias"]))
            cse = torch.sigmoid(F.conv2d(s, W[f"{name}.attention2.cSE.3.weight"], W[f"{name}.attention2.cSE.3.bias"]))
            sse = torch.sigmoid(F.conv2d(x, W[f"{name}.attention2.sSE.0.weight"], W[f"{name}.attention2.sSE.0.bias"]))
            x = x * cse + x * sse
            _tap(taps, f"{name}.scse", x)
        return x

    decx8 = block(x16, encx8, "decoder.upsampler1", False, True)
    decx4 = block(decx8, encx4, "decoder.upsampler2", True, False)
    return decx4


def heatmap_decode(hm: torch.Tensor):
    """COTRAIN.postp (model.py:511-554): returns (loc_fix [B,196], score [B,98], idx [B,98])."""
    bs, _, h, w = hm.shape
    flat = hm.reshape(bs, 3, NUM_POINTS, h * w)
    score, idx = torch.max(flat[:, 0], dim=2)
    ox = torch.gather(flat[:, 1], 2, idx.unsqueeze(-1)).squeeze(-1)
    oy = torch.gather(flat[:, 2], 2, idx.unsqueeze(-1)).squeeze(-1)
    xs = ((idx % w) + ox) / w
    ys = ((idx // w) + oy) / h
    loc_fix = torch.stack([xs, ys], 2).to(hm.dtype).reshape(bs, -1)
    return loc_fix, score, idx


def student_forward(W: Dict[str, torch.Tensor], x: torch.Tensor, taps=None):
    """COTRAIN(inference='student').forward (model.py:556-568) == the exported ONNX graph
    (tools/convert_to_onnx.py:28,54-61).  ``x``: [B,3,S,S] float, the caller's channel order
    (BGR/255 in the pipeline, face_landmark.py:44-47).  Returns (loc_fix [B,196], score [B,98]).
    The pose/cls head (model.py:286-293) is dead at inference and is not evaluated."""
    feats = encoder_forward(W, x, taps)
    decx4 = decoder_forward(W, feats, taps)
    hm = F.conv2d(decx4, W["hm.weight"], W["hm.bias"])
    _tap(taps, "hm", hm)
    loc_fix, score, idx = heatmap_decode(hm)
    _tap(taps, "hm_idx", idx)
    return loc_fix, score


# --------------------------------------------------------------------------------------
# cost model cross-check against README.md:34-37 ("Flops(G)" = thop MACs / 1024^3, model.py:594-601)
# --------------------------------------------------------------------------------------
def count_macs_params(size: int = 256) -> Tuple[int, int]:
    """Analytic conv/linear MACs and parameter count of the inference graph + dead fc."""
    macs = 0
    params = 0
    hw = {}

    def conv(cout, cin_g, k, h, w, bias=False):
        nonlocal macs, params
        macs += cout * cin_g * k * k * h * w
        params += cout * cin_g * k * k + (cout if bias else 0)

    def bn(ch):
        nonlocal params
        params += 2 * ch

    s = size // 2
    conv(16, 3, 3, s, s); bn(16)
    for stage in encoder_blocks():
        for b in stage:
            so = s // b.stride
            if b.kind == "ds":
                conv(b.mid, 1, b.k, so, so); bn(b.mid)
                conv(b.cout, b.mid, 1, so, so); bn(b.cout)
            else:
                conv(b.mid, b.cin, 1, s, s); bn(b.mid)
                conv(b.mid, 1, b.k, so, so); bn(b.mid)
                if b.se_rd:
                    conv(b.se_rd, b.mid, 1, 1, 1, True); conv(b.mid, b.se_rd, 1, 1, 1, True)
                conv(b.cout, b.mid, 1, so, so); bn(b.cout)
            s = so
    h16 = size // 16
    conv(64, 160, 1, h16, h16); conv(64, 160, 3, h16, h16); conv(64, 160, 3, h16, h16)
    conv(64, 160, 1, 1, 1); bn(64); bn(256)
    conv(256, 256, 1, h16, h16); bn(256)
    h8, h4 = size // 8, size // 4
    conv(296, 1, 3, h8, h8, True); bn(296); conv(256, 296, 1, h8, h8); bn(256)
    conv(64, 256, 1, 1, 1, True); conv(256, 64, 1, 1, 1, True); conv(1, 256, 1, h8, h8, True)
    conv(280, 1, 3, h4, h4, True); bn(280); conv(128, 280, 1, h4, h4); bn(128)
    conv(128, 128, 3, h4, h4, True); bn(128)
    conv(294, 128, 1, h4, h4, True)
    params += 640 * 7 + 7
    return macs, params


def to_torch(weights_np: Dict[str, "object"], dtype=torch.float32) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(v).to(dtype) for k, v in weights_np.items()}
