"""Random-init weights of the Student landmark regressor (for benchmarks / plumbing checks).

The reference's trained blobs are not in the checkout and there is no network, so throughput is
measured on random weights of the exact architecture (BASELINE contract: "random-init weights of
that architecture").  Convs follow the reference's ``weight_init`` (kaiming-normal fan_out,
model.py:199-209); BatchNorm statistics are set analytically so activations stay O(1) through
all ~60 layers (running_var = expected output variance of the conv in front of it).
Parity tests do NOT use this module: they use the oracle's calibrated synthetic weights.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from .student import _STAGES, NUM_POINTS


def _make_divisible(v: float, divisor: int = 8, round_limit: float = 0.9) -> int:
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < round_limit * v:
        new_v += divisor
    return new_v


def student_param_shapes() -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, kind) with kind in conv / bias / bn (a BN contributes 4 tensors)."""
    out: List[Tuple[str, Tuple[int, ...], str]] = [("encoder.conv_stem.weight", (16, 3, 3, 3), "conv"),
                                                    ("encoder.bn1", (16,), "bn")]
    cin = 16
    for si, stack in enumerate(_STAGES):
        for bi, (kind, k, s, e, cout, se, act) in enumerate(stack):
            p = f"encoder.blocks.{si}.{bi}"
            if kind == "ds":
                out += [(f"{p}.conv_dw.weight", (cin, 1, k, k), "conv"), (f"{p}.bn1", (cin,), "bn"),
                        (f"{p}.conv_pw.weight", (cout, cin, 1, 1), "conv"), (f"{p}.bn2", (cout,), "bn")]
            else:
                mid = _make_divisible(cin * e)
                out += [(f"{p}.conv_pw.weight", (mid, cin, 1, 1), "conv"), (f"{p}.bn1", (mid,), "bn"),
                        (f"{p}.conv_dw.weight", (mid, 1, k, k), "conv"), (f"{p}.bn2", (mid,), "bn")]
                if se:
                    rd = _make_divisible(mid * 0.25)
                    out += [(f"{p}.se.conv_reduce.weight", (rd, mid, 1, 1), "conv"), (f"{p}.se.conv_reduce.bias", (rd,), "bias"),
                            (f"{p}.se.conv_expand.weight", (mid, rd, 1, 1), "conv"), (f"{p}.se.conv_expand.bias", (mid,), "bias")]
                out += [(f"{p}.conv_pwl.weight", (cout, mid, 1, 1), "conv"), (f"{p}.bn3", (cout,), "bn")]
            cin = cout
    a = "decoder.aspp"
    out += [(f"{a}.conv1.weight", (64, 160, 1, 1), "conv"), (f"{a}.conv2.weight", (64, 160, 3, 3), "conv"),
            (f"{a}.conv3.weight", (64, 160, 3, 3), "conv"), (f"{a}.bn_act.0", (256,), "bn"),
            (f"{a}.fm_pool.pool.1.weight", (64, 160, 1, 1), "conv"), (f"{a}.fm_pool.pool.2", (64,), "bn"),
            (f"{a}.project.0.weight", (256, 256, 1, 1), "conv"), (f"{a}.project.1", (256,), "bn")]
    for name, c_in, c_out, second, att in (("decoder.upsampler1", 296, 256, False, True),
                                           ("decoder.upsampler2", 280, 128, True, False)):
        out += [(f"{name}.conv1.0.conv_dw.0.weight", (c_in, 1, 3, 3), "conv"), (f"{name}.conv1.0.conv_dw.0.bias", (c_in,), "bias"),
                (f"{name}.conv1.0.conv_dw.1", (c_in,), "bn"), (f"{name}.conv1.0.conv_pw.weight", (c_out, c_in, 1, 1), "conv"),
                (f"{name}.conv1.1", (c_out,), "bn")]
        if second:
            out += [(f"{name}.conv2.0.weight", (c_out, c_out, 3, 3), "conv"), (f"{name}.conv2.0.bias", (c_out,), "bias"),
                    (f"{name}.conv2.1", (c_out,), "bn")]
        if att:
            out += [(f"{name}.attention2.cSE.1.weight", (c_out // 4, c_out, 1, 1), "conv"), (f"{name}.attention2.cSE.1.bias", (c_out // 4,), "bias"),
                    (f"{name}.attention2.cSE.3.weight", (c_out, c_out // 4, 1, 1), "conv"), (f"{name}.attention2.cSE.3.bias", (c_out,), "bias"),
                    (f"{name}.attention2.sSE.0.weight", (1, c_out, 1, 1), "conv"), (f"{name}.attention2.sSE.0.bias", (1,), "bias")]
    out += [("hm.weight", (NUM_POINTS * 3, 128, 1, 1), "conv"), ("hm.bias", (NUM_POINTS * 3,), "bias")]
    return out


def random_student_weights(seed: int = 0) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    w: Dict[str, np.ndarray] = {}
    last_conv_var = 1.0
    for name, shape, kind in student_param_shapes():
        if kind == "conv":
            cout, cin_g, kh, kw = shape
            std = np.sqrt(2.0 / (cout * kh * kw))
            w[name] = (rng.standard_normal(shape) * std).astype(np.float32)
            last_conv_var = max(cin_g * kh * kw * std * std * 0.6, 1e-3)  # E[x^2] ~ 0.6 after ReLU-like acts
        elif kind == "bias":
            w[name] = (rng.standard_normal(shape) * 0.05).astype(np.float32)
        else:
            w[f"{name}.weight"] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
            w[f"{name}.bias"] = (rng.standard_normal(shape) * 0.3).astype(np.float32)
            w[f"{name}.running_mean"] = np.zeros(shape, np.float32)
            w[f"{name}.running_var"] = np.full(shape, last_conv_var, np.float32)
    return w
