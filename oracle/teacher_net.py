"""Oracle: torch-CPU restatement of the Teacher landmark regressor.  TEST INFRASTRUCTURE ONLY.

``TeacherNet`` (TRAIN/face_landmark/lib/core/base_trainer/model.py:302-345) = timm ``hrnet_w18``
features (``features_only=True, out_indices=[0,1,2,3]``, model.py:306-311) + the same ``Decoder`` /
``hm`` head / ``postp`` as the student with ``encoder.out_channels = [3,64,128,256,512]`` (:313-315).

* decoder / heads / decode: reuse ``oracle.landmark_net`` pieces, PINNED against the reference's own
  ``TeacherNet`` classes: ``oracle.ref_import.load_reference_cotrain(student, teacher, inference='teacher')``
  executed live (``tests/test_oracle_pinned.py::test_oracle_equals_reference_live``, bit-identical) and through
  the committed golden ``tests/golden/landmark_teacher128.npz`` (``::test_teacher_oracle_reproduces_reference_golden``).
* encoder: timm==0.6.11 ``HighResolutionNetFeatures`` (feature_location='incre') is NOT vendored in the
  reference: restated from the published architecture -- PARITY UNPINNED.
  hrnet_w18: stem conv3x3 s2 (3->64) [feature 0] -> conv3x3 s2 (64->64) -> layer1 = 4 Bottlenecks (->256)
  -> transition1 -> stage2 (1 module, 2 branches 18/36, 4 BasicBlocks each) -> transition2 ->
  stage3 (4 modules, 3 branches 18/36/72) -> transition3 -> stage4 (3 modules, 4 branches 18/36/72/144)
  -> one "incre" Bottleneck per branch (-> 128/256/512/[1024]) [features 1,2,3; the /32 head is unused].
  State-dict names follow timm (``encoder.stage3.1.fuse_layers.0.2.0.weight`` ...).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

from . import landmark_net as ln

BN_EPS = 1e-5
BRANCH_CH = [18, 36, 72, 144]
STAGES = [(2, 1, 2), (3, 4, 3), (4, 3, 4)]  # (stage index, modules, branches)
HEAD_PLANES = [32, 64, 128, 256]
ENCODER_OUT_CHANNELS = [64, 128, 256, 512]

_CALIBRATING = None


def _bn_entries(prefix, ch):
    return [(f"{prefix}.weight", (ch,), "bn_gamma"), (f"{prefix}.bias", (ch,), "bn_beta"),
            (f"{prefix}.running_mean", (ch,), "bn_mean"), (f"{prefix}.running_var", (ch,), "bn_var")]


def _conv_bn(prefix_conv, prefix_bn, cin, cout, k):
    return [(f"{prefix_conv}.weight", (cout, cin, k, k), "conv")] + _bn_entries(prefix_bn, cout)


def _basic_entries(p, ch):
    return _conv_bn(f"{p}.conv1", f"{p}.bn1", ch, ch, 3) + _conv_bn(f"{p}.conv2", f"{p}.bn2", ch, ch, 3)


def _bottleneck_entries(p, cin, planes):
    out = (_conv_bn(f"{p}.conv1", f"{p}.bn1", cin, planes, 1) + _conv_bn(f"{p}.conv2", f"{p}.bn2", planes, planes, 3) +
           _conv_bn(f"{p}.conv3", f"{p}.bn3", planes, planes * 4, 1))
    if cin != planes * 4:
        out += _conv_bn(f"{p}.downsample.0", f"{p}.downsample.1", cin, planes * 4, 1)
    return out


def encoder_inventory() -> List[Tuple[str, Tuple[int, ...], str]]:
    e = "encoder"
    inv = _conv_bn(f"{e}.conv1", f"{e}.bn1", 3, 64, 3) + _conv_bn(f"{e}.conv2", f"{e}.bn2", 64, 64, 3)
    cin = 64
    for b in range(4):
        inv += _bottleneck_entries(f"{e}.layer1.{b}", cin, 64)
        cin = 256
    inv += _conv_bn(f"{e}.transition1.0.0", f"{e}.transition1.0.1", 256, 18, 3)
    inv += _conv_bn(f"{e}.transition1.1.0.0", f"{e}.transition1.1.0.1", 256, 36, 3)
    for si, modules, nb in STAGES:
        if si > 2:
            inv += _conv_bn(f"{e}.transition{si - 1}.{nb - 1}.0.0", f"{e}.transition{si - 1}.{nb - 1}.0.1",
                            BRANCH_CH[nb - 2], BRANCH_CH[nb - 1], 3)
        for m in range(modules):
            p = f"{e}.stage{si}.{m}"
            for br in range(nb):
                for blk in range(4):
                    inv += _basic_entries(f"{p}.branches.{br}.{blk}", BRANCH_CH[br])
            for i in range(nb):
                for j in range(nb):
                    if j > i:
                        inv += _conv_bn(f"{p}.fuse_layers.{i}.{j}.0", f"{p}.fuse_layers.{i}.{j}.1", BRANCH_CH[j], BRANCH_CH[i], 1)
                    elif j < i:
                        for k in range(i - j):
                            last = k == i - j - 1
                            cout = BRANCH_CH[i] if last else BRANCH_CH[j]
                            inv += _conv_bn(f"{p}.fuse_layers.{i}.{j}.{k}.0", f"{p}.fuse_layers.{i}.{j}.{k}.1", BRANCH_CH[j], cout, 3)
    for i in range(3):  # the 4th incre module (1024 @ /32) is not requested by out_indices=[0,1,2,3]
        inv += _bottleneck_entries(f"{e}.incre_modules.{i}.0", BRANCH_CH[i], HEAD_PLANES[i])
    return inv


def param_inventory() -> List[Tuple[str, Tuple[int, ...], str]]:
    """All tensors of ``COTRAIN.teacher`` used at inference (names relative to it)."""
    inv = encoder_inventory()
    c16, c8, c4 = ENCODER_OUT_CHANNELS[3], ENCODER_OUT_CHANNELS[2], ENCODER_OUT_CHANNELS[1]
    a = "decoder.aspp"
    inv += [(f"{a}.conv1.weight", (64, c16, 1, 1), "conv"), (f"{a}.conv2.weight", (64, c16, 3, 3), "conv"),
            (f"{a}.conv3.weight", (64, c16, 3, 3), "conv")]
    inv += _bn_entries(f"{a}.bn_act.0", 256)
    inv += [(f"{a}.fm_pool.pool.1.weight", (64, c16, 1, 1), "conv")] + _bn_entries(f"{a}.fm_pool.pool.2", 64)
    inv += [(f"{a}.project.0.weight", (256, 256, 1, 1), "conv")] + _bn_entries(f"{a}.project.1", 256)
    for name, cin, cout, second, att in (("decoder.upsampler1", 256 + c8, 256, False, True),
                                         ("decoder.upsampler2", 256 + c4, 128, True, False)):
        inv += [(f"{name}.conv1.0.conv_dw.0.weight", (cin, 1, 3, 3), "conv"), (f"{name}.conv1.0.conv_dw.0.bias", (cin,), "bias")]
        inv += _bn_entries(f"{name}.conv1.0.conv_dw.1", cin)
        inv += [(f"{name}.conv1.0.conv_pw.weight", (cout, cin, 1, 1), "conv")] + _bn_entries(f"{name}.conv1.1", cout)
        if second:
            inv += [(f"{name}.conv2.0.weight", (cout, cout, 3, 3), "conv"), (f"{name}.conv2.0.bias", (cout,), "bias")]
            inv += _bn_entries(f"{name}.conv2.1", cout)
        if att:
            inv += [(f"{name}.attention2.cSE.1.weight", (cout // 4, cout, 1, 1), "conv"), (f"{name}.attention2.cSE.1.bias", (cout // 4,), "bias"),
                    (f"{name}.attention2.cSE.3.weight", (cout, cout // 4, 1, 1), "conv"), (f"{name}.attention2.cSE.3.bias", (cout,), "bias"),
                    (f"{name}.attention2.sSE.0.weight", (1, cout, 1, 1), "conv"), (f"{name}.attention2.sSE.0.bias", (1,), "bias")]
    inv += [("fc.weight", (7, 640), "linear"), ("fc.bias", (7,), "bias")]
    inv += [("hm.weight", (ln.NUM_POINTS * 3, 128, 1, 1), "conv"), ("hm.bias", (ln.NUM_POINTS * 3,), "bias")]
    return inv


def _bn(W, prefix, x):
    if _CALIBRATING is not None:
        W[f"{prefix}.running_mean"] = x.mean((0, 2, 3)).float().to(x.dtype)
        W[f"{prefix}.running_var"] = x.var((0, 2, 3), unbiased=False).float().clamp_min(1e-3).to(x.dtype)
        _CALIBRATING.append(prefix)
    return F.batch_norm(x, W[f"{prefix}.running_mean"], W[f"{prefix}.running_var"],
                        W[f"{prefix}.weight"], W[f"{prefix}.bias"], False, 0.0, BN_EPS)


def _cb(W, pc, pb, x, stride=1, relu=True):
    k = W[f"{pc}.weight"].shape[-1]
    y = _bn(W, pb, F.conv2d(x, W[f"{pc}.weight"], None, stride, k // 2))
    return F.relu(y) if relu else y


def _basic(W, p, x):
    y = _cb(W, f"{p}.conv1", f"{p}.bn1", x)
    y = _cb(W, f"{p}.conv2", f"{p}.bn2", y, relu=False)
    return F.relu(y + x)


def _bottleneck(W, p, x):
    y = _cb(W, f"{p}.conv1", f"{p}.bn1", x)
    y = _cb(W, f"{p}.conv2", f"{p}.bn2", y)
    y = _cb(W, f"{p}.conv3", f"{p}.bn3", y, relu=False)
    sc = _cb(W, f"{p}.downsample.0", f"{p}.downsample.1", x, relu=False) if f"{p}.downsample.0.weight" in W else x
    return F.relu(y + sc)


def _hr_module(W, p, xs: List[torch.Tensor]) -> List[torch.Tensor]:
    nb = len(xs)
    xs = list(xs)
    for br in range(nb):
        for blk in range(4):
            xs[br] = _basic(W, f"{p}.branches.{br}.{blk}", xs[br])
    out = []
    for i in range(nb):
        y = None
        for j in range(nb):
            if j == i:
                t = xs[j]
            elif j > i:
                t = _cb(W, f"{p}.fuse_layers.{i}.{j}.0", f"{p}.fuse_layers.{i}.{j}.1", xs[j], relu=False)
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode="nearest")
            else:
                t = xs[j]
                for k in range(i - j):
                    t = _cb(W, f"{p}.fuse_layers.{i}.{j}.{k}.0", f"{p}.fuse_layers.{i}.{j}.{k}.1", t, stride=2,
                            relu=(k != i - j - 1))
            y = t if y is None else y + t
        out.append(F.relu(y))
    return out


def encoder_forward(W: Dict[str, torch.Tensor], x: torch.Tensor, taps=None) -> List[torch.Tensor]:
    """timm HighResolutionNetFeatures.forward -> [64@/2, 128@/4, 256@/8, 512@/16].  UNPINNED restatement."""
    def tap(n, t):
        if taps is not None:
            taps[n] = t
    e = "encoder"
    x = _cb(W, f"{e}.conv1", f"{e}.bn1", x, stride=2)
    f0 = x
    tap("encoder.stem", x)
    x = _cb(W, f"{e}.conv2", f"{e}.bn2", x, stride=2)
    for b in range(4):
        x = _bottleneck(W, f"{e}.layer1.{b}", x)
    tap("encoder.layer1", x)
    xs = [_cb(W, f"{e}.transition1.0.0", f"{e}.transition1.0.1", x),
          _cb(W, f"{e}.transition1.1.0.0", f"{e}.transition1.1.0.1", x, stride=2)]
    for si, modules, nb in STAGES:
        if si > 2:
            xs = xs + [_cb(W, f"{e}.transition{si - 1}.{nb - 1}.0.0", f"{e}.transition{si - 1}.{nb - 1}.0.1", xs[-1], stride=2)]
        for m in range(modules):
            xs = _hr_module(W, f"{e}.stage{si}.{m}", xs)
        for i, t in enumerate(xs):
            tap(f"encoder.stage{si}.branch{i}", t)
    feats = [f0]
    for i in range(3):
        feats.append(_bottleneck(W, f"{e}.incre_modules.{i}.0", xs[i]))
        tap(f"encoder.incre{i}", feats[-1])
    return feats


def teacher_forward(W: Dict[str, torch.Tensor], x: torch.Tensor, taps=None):
    """COTRAIN(inference='teacher').forward (model.py:556-568): (loc_fix [B,196], score [B,98])."""
    feats = encoder_forward(W, x, taps)
    decx4 = ln.decoder_forward(W, feats, taps)
    hm = F.conv2d(decx4, W["hm.weight"], W["hm.bias"])
    if taps is not None:
        taps["hm"] = hm
    loc_fix, score, idx = ln.heatmap_decode(hm)
    if taps is not None:
        taps["hm_idx"] = idx
    return loc_fix, score
