"""Teacher landmark regressor (HRNet-W18 encoder) -> packed HIP program.

Graph source: ``TeacherNet`` (TRAIN/face_landmark/lib/core/base_trainer/model.py:302-345): timm
``hrnet_w18`` features (out_indices [0,1,2,3], model.py:306-311) + the shared ``Decoder`` / ``hm`` head /
``postp`` (see ``graph/student.py::build_decoder_and_head``).  BASELINE config 5.

HRNet-W18 (timm 0.6.11 ``HighResolutionNetFeatures``, feature_location='incre'; not vendored in the
reference, restated): stem 3x3 s2 (feature /2) -> 3x3 s2 -> layer1 (4 Bottlenecks) -> 3 multi-branch
stages (18/36/72/144 channels at /4../32) with fuse layers -> one Bottleneck "incre" head per branch
(128/256/512 channels at /4,/8,/16; the /32 head is not requested and never built).
``weights``: ``{name: ndarray}`` with ``COTRAIN.teacher`` state_dict names.

Channel counts that are not a multiple of the 16-byte vector (18 -> 20 f32 / 24 f16) are padded; the conv
epilogue writes the padding as zeros so downstream vector kernels stay exact.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np

from . import ir
from .student import build_decoder_and_head

BRANCH_CH = [18, 36, 72, 144]
STAGES = [(2, 1, 2), (3, 4, 3), (4, 3, 4)]  # (stage index, modules, branches)
# model.py:313 keeps features [stem, incre(branch 0), incre(branch 1), incre(branch 2)]: the fourth (144-channel) OUTPUT of
# the last HighResolutionModule feeds nothing.  Its six fuse convolutions are dead code -- an ONNX export of the model does
# not even contain them (the exporter drops nodes that reach no output) -- so the program skips them and the ONNX importer
# does not expect them.
DEAD_FUSE_OUTPUT = (4, 2, 3)                       # (stage, module, output branch)
DEAD_FUSE_PREFIX = "encoder.stage4.2.fuse_layers.3."


def _bn(w, prefix):
    return {k: w[f"{prefix}.{k}"] for k in ("weight", "bias", "running_mean", "running_var")}


def build_teacher_program(weights: Dict[str, np.ndarray], input_size: int = 256, dtype: str = "f32s",
                          keep_all: bool = False, debug_full_hm: bool = False, fuse_chains: bool = True, one_product=()):
    """``one_product`` (opt-in, f32s): decoder layers on ONE f16 product, as in ``build_student_program`` -- "hero" and "head" are two of the
    four groups the per-layer study found tolerant on the Teacher (profiles/r06_teacher_precision_study.txt: 1.2e-4 and 1.0e-4 alone)."""
    assert input_size % 64 == 0
    w = weights
    pb = ir.ProgramBuilder(dtype, input_size, input_size, keep_all=keep_all)

    def cb(x, pc, pbn, act, stride=1, res=-1, name=""):
        wt, b = ir.fold_bn(w[f"{pc}.weight"], None, _bn(w, pbn))
        k = wt.shape[-1]
        return pb.conv(x, wt, b, act, stride=stride, pad=k // 2, res=res, out_name=name)

    def basic(x, p):
        y = cb(x, f"{p}.conv1", f"{p}.bn1", "relu")
        return cb(y, f"{p}.conv2", f"{p}.bn2", "relu", res=x)          # relu(bn2(conv2) + x)

    def basic_folded(p):
        w1, b1 = ir.fold_bn(w[f"{p}.conv1.weight"], None, _bn(w, f"{p}.bn1"))
        w2, b2 = ir.fold_bn(w[f"{p}.conv2.weight"], None, _bn(w, f"{p}.bn2"))
        return w1, b1, w2, b2

    def bottleneck(x, p, name=""):
        has_ds = f"{p}.downsample.0.weight" in w
        if fuse_chains and pb.hr_bottleneck_supported(x, w[f"{p}.conv1.weight"].shape[0], w[f"{p}.conv3.weight"].shape[0], has_ds):
            f = lambda c, bn: ir.fold_bn(w[f"{p}.{c}.weight"], None, _bn(w, f"{p}.{bn}"))     # the whole block in one launch (csrc/k_hrb.h)
            ds = f("downsample.0", "downsample.1") if has_ds else (None, None)
            return pb.hr_bottleneck(x, *f("conv1", "bn1"), *f("conv2", "bn2"), *f("conv3", "bn3"), *ds, out_name=name)
        sc = cb(x, f"{p}.downsample.0", f"{p}.downsample.1", "none") if f"{p}.downsample.0.weight" in w else x
        y = cb(x, f"{p}.conv1", f"{p}.bn1", "relu")
        y = cb(y, f"{p}.conv2", f"{p}.bn2", "relu")
        return cb(y, f"{p}.conv3", f"{p}.bn3", "relu", res=sc, name=name)

    e = "encoder"
    wt, b = ir.fold_bn(w[f"{e}.conv1.weight"], None, _bn(w, f"{e}.bn1"))
    f0 = pb.stem(wt, b, "relu", out_name="encoder.stem")   # 3 -> 64, 3x3 stride 2 (feature /2, unused by the decoder)
    x = cb(f0, f"{e}.conv2", f"{e}.bn2", "relu", stride=2)
    for blk in range(4):
        x = bottleneck(x, f"{e}.layer1.{blk}", name="encoder.layer1" if blk == 3 else "")
    xs: List[int] = [cb(x, f"{e}.transition1.0.0", f"{e}.transition1.0.1", "relu"),
                     cb(x, f"{e}.transition1.1.0.0", f"{e}.transition1.1.0.1", "relu", stride=2)]
    for si, modules, nb in STAGES:
        if si > 2:
            xs = xs + [cb(xs[-1], f"{e}.transition{si - 1}.{nb - 1}.0.0", f"{e}.transition{si - 1}.{nb - 1}.0.1", "relu", stride=2)]
        for m in range(modules):
            p = f"{e}.stage{si}.{m}"
            for br in range(nb):
                if fuse_chains and pb.basic_chain_supported(xs[br], 4):      # the branch's four blocks in one launch
                    xs[br] = pb.basic_chain(xs[br], [basic_folded(f"{p}.branches.{br}.{blk}") for blk in range(4)])
                    continue
                for blk in range(4):
                    if fuse_chains and pb.basic_block_supported(xs[br]):               # both convs of the block in one launch
                        xs[br] = pb.basic_block(xs[br], *basic_folded(f"{p}.branches.{br}.{blk}"))
                    else:
                        xs[br] = basic(xs[br], f"{p}.branches.{br}.{blk}")
            fused = []
            for i in range(nb):
                if (si, m, i) == DEAD_FUSE_OUTPUT:
                    continue           # nothing reads the 144-channel output of the very last module (see DEAD_FUSE_PREFIX)
                last_stage_module = (m == modules - 1)
                name = f"encoder.stage{si}.branch{i}" if last_stage_module else ""
                down = [j for j in range(nb) if j < i]
                up = [j for j in range(nb) if j > i]
                y = xs[i]
                n_terms = len(down) + len(up)
                done = 0
                for j in down:     # strided 3x3 chains; the last conv of each chain accumulates into y
                    t = xs[j]
                    for k in range(i - j):
                        lastc = k == i - j - 1
                        done += 1 if lastc else 0
                        final = lastc and done == n_terms
                        t = cb(t, f"{p}.fuse_layers.{i}.{j}.{k}.0", f"{p}.fuse_layers.{i}.{j}.{k}.1",
                               ("relu" if final else "none") if lastc else "relu", stride=2, res=y if lastc else -1,
                               name=name if final else "")
                    y = t
                if up and fuse_chains and pb.fuse_up_supported(y, [(xs[j], j - i) for j in up]):
                    # every upsampled term of this output in one launch (csrc/k_layers.h fuse_up_kernel); they are the last terms of the sum
                    terms = [(xs[j], *ir.fold_bn(w[f"{p}.fuse_layers.{i}.{j}.0.weight"], None, _bn(w, f"{p}.fuse_layers.{i}.{j}.1")), j - i) for j in up]
                    y = pb.fuse_up(y, terms, "relu", out_name=name)
                    up = []
                for j in up:       # 1x1 conv at low resolution, nearest upsample, add
                    t = cb(xs[j], f"{p}.fuse_layers.{i}.{j}.0", f"{p}.fuse_layers.{i}.{j}.1", "none")
                    done += 1
                    final = done == n_terms
                    y = pb.add_up(y, t, j - i, "relu" if final else "none", out_name=name if final else "")
                fused.append(y)
            xs = fused
    feats = [bottleneck(xs[i], f"{e}.incre_modules.{i}.0", name=f"encoder.incre{i}") for i in range(3)]
    loc, score, info = build_decoder_and_head(pb, w, feats[0], feats[1], feats[2], input_size, keep_all, debug_full_hm, tuple(one_product))
    blob = pb.finish([loc, score])
    info.update({"tensors": dict(pb.tensor_names), "input_size": input_size, "dtype": dtype,
                 "n_ops": len(pb.ops), "const_bytes": len(pb.consts)})
    return blob, info


def teacher_param_shapes():
    """(name, shape, kind) of every tensor of ``COTRAIN.teacher`` used at inference (kind: conv/bias/bn)."""
    out = []

    def cb(pc, pbn, cin, cout, k):
        out.extend([(f"{pc}.weight", (cout, cin, k, k), "conv"), (pbn, (cout,), "bn")])

    def bottleneck(p, cin, planes):
        cb(f"{p}.conv1", f"{p}.bn1", cin, planes, 1); cb(f"{p}.conv2", f"{p}.bn2", planes, planes, 3)
        cb(f"{p}.conv3", f"{p}.bn3", planes, planes * 4, 1)
        if cin != planes * 4:
            cb(f"{p}.downsample.0", f"{p}.downsample.1", cin, planes * 4, 1)

    e = "encoder"
    cb(f"{e}.conv1", f"{e}.bn1", 3, 64, 3); cb(f"{e}.conv2", f"{e}.bn2", 64, 64, 3)
    cin = 64
    for b in range(4):
        bottleneck(f"{e}.layer1.{b}", cin, 64)
        cin = 256
    cb(f"{e}.transition1.0.0", f"{e}.transition1.0.1", 256, 18, 3)
    cb(f"{e}.transition1.1.0.0", f"{e}.transition1.1.0.1", 256, 36, 3)
    for si, modules, nb in STAGES:
        if si > 2:
            cb(f"{e}.transition{si - 1}.{nb - 1}.0.0", f"{e}.transition{si - 1}.{nb - 1}.0.1", BRANCH_CH[nb - 2], BRANCH_CH[nb - 1], 3)
        for m in range(modules):
            p = f"{e}.stage{si}.{m}"
            for br in range(nb):
                for blk in range(4):
                    q = f"{p}.branches.{br}.{blk}"
                    cb(f"{q}.conv1", f"{q}.bn1", BRANCH_CH[br], BRANCH_CH[br], 3)
                    cb(f"{q}.conv2", f"{q}.bn2", BRANCH_CH[br], BRANCH_CH[br], 3)
            for i in range(nb):
                for j in range(nb):
                    if j > i:
                        cb(f"{p}.fuse_layers.{i}.{j}.0", f"{p}.fuse_layers.{i}.{j}.1", BRANCH_CH[j], BRANCH_CH[i], 1)
                    elif j < i:
                        for k in range(i - j):
                            cout = BRANCH_CH[i] if k == i - j - 1 else BRANCH_CH[j]
                            cb(f"{p}.fuse_layers.{i}.{j}.{k}.0", f"{p}.fuse_layers.{i}.{j}.{k}.1", BRANCH_CH[j], cout, 3)
    for i, planes in enumerate((32, 64, 128)):
        bottleneck(f"{e}.incre_modules.{i}.0", BRANCH_CH[i], planes)
    a = "decoder.aspp"
    out.extend([(f"{a}.conv1.weight", (64, 512, 1, 1), "conv"), (f"{a}.conv2.weight", (64, 512, 3, 3), "conv"),
                (f"{a}.conv3.weight", (64, 512, 3, 3), "conv"), (f"{a}.bn_act.0", (256,), "bn"),
                (f"{a}.fm_pool.pool.1.weight", (64, 512, 1, 1), "conv"), (f"{a}.fm_pool.pool.2", (64,), "bn"),
                (f"{a}.project.0.weight", (256, 256, 1, 1), "conv"), (f"{a}.project.1", (256,), "bn")])
    for name, c_in, c_out, second, att in (("decoder.upsampler1", 512, 256, False, True), ("decoder.upsampler2", 384, 128, True, False)):
        out.extend([(f"{name}.conv1.0.conv_dw.0.weight", (c_in, 1, 3, 3), "conv"), (f"{name}.conv1.0.conv_dw.0.bias", (c_in,), "bias"),
                    (f"{name}.conv1.0.conv_dw.1", (c_in,), "bn"), (f"{name}.conv1.0.conv_pw.weight", (c_out, c_in, 1, 1), "conv"),
                    (f"{name}.conv1.1", (c_out,), "bn")])
        if second:
            out.extend([(f"{name}.conv2.0.weight", (c_out, c_out, 3, 3), "conv"), (f"{name}.conv2.0.bias", (c_out,), "bias"),
                        (f"{name}.conv2.1", (c_out,), "bn")])
        if att:
            out.extend([(f"{name}.attention2.cSE.1.weight", (c_out // 4, c_out, 1, 1), "conv"), (f"{name}.attention2.cSE.1.bias", (c_out // 4,), "bias"),
                        (f"{name}.attention2.cSE.3.weight", (c_out, c_out // 4, 1, 1), "conv"), (f"{name}.attention2.cSE.3.bias", (c_out,), "bias"),
                        (f"{name}.attention2.sSE.0.weight", (1, c_out, 1, 1), "conv"), (f"{name}.attention2.sSE.0.bias", (1,), "bias")])
    out.extend([("hm.weight", (294, 128, 1, 1), "conv"), ("hm.bias", (294,), "bias")])
    return out


def random_teacher_weights(seed: int = 2) -> Dict[str, np.ndarray]:
    """Random-init weights of the exact architecture (benchmarks only; see graph/random_init.py)."""
    rng = np.random.default_rng(seed)
    w: Dict[str, np.ndarray] = {}
    last_var = 1.0
    for name, shape, kind in teacher_param_shapes():
        if kind == "conv":
            cout, cin_g, kh, kw = shape
            std = np.sqrt(2.0 / (cout * kh * kw))
            w[name] = (rng.standard_normal(shape) * std).astype(np.float32)
            last_var = max(cin_g * kh * kw * std * std * 0.6, 1e-3)
        elif kind == "bias":
            w[name] = (rng.standard_normal(shape) * 0.05).astype(np.float32)
        else:
            # the BatchNorms that close a residual branch (BasicBlock.bn2 / Bottleneck.bn3) or feed a fuse sum get a small gain: HRNet stacks
            # ~40 residual blocks and with unit gains the activations of a random net grow past 6e4 (the range guard of the
            # f32s kernels fires, correctly); trained nets do not do that
            damp = name.endswith((".bn2", ".bn3")) or "fuse_layers" in name
            g_lo, g_hi = (0.08, 0.25) if damp else (0.5, 1.5)
            w[f"{name}.weight"] = rng.uniform(g_lo, g_hi, shape).astype(np.float32)
            w[f"{name}.bias"] = (rng.standard_normal(shape) * 0.3).astype(np.float32)
            w[f"{name}.running_mean"] = np.zeros(shape, np.float32)
            w[f"{name}.running_var"] = np.full(shape, last_var, np.float32)
    return w
