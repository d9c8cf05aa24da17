"""Which Teacher convolutions tolerate ONE f16 product?  (round-5 VERDICT item 6b; BASELINE configs[4] names "fp16 MFMA")

CPU study on the oracle (TEST INFRASTRUCTURE: imports oracle/): a dense convolution that the engine would run as a single
v_mfma_f32_16x16x32_f16 product per 32 k -- both operands rounded to f16, exact products, f32 accumulation -- is simulated by
rounding that convolution's input and weight tensors to f16 and running it in f32.  f32s (three products) is represented by the
unrounded convolution (its error, 4e-7 of the range per layer, is far below what is measured here).  Groups of layers are switched
one at a time, then cumulatively in order of harmlessness; the figure of merit is the one the GPU parity tests use: max
|loc - oracle| over the landmarks whose oracle top-1 / top-2 heat-map margin is above 2e-3, in normalised crop coordinates
(north star: 1e-3; budget for a precision mix: 2.5e-4).

    python tools/teacher_precision_study.py [--faces 4] [--out profiles/r06_teacher_precision_study.json]
"""
from __future__ import annotations

import argparse
import json
import os
import re
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import landmark_net as ln          # noqa: E402
from oracle import synth_weights as sw         # noqa: E402
from oracle import teacher_net as tn           # noqa: E402


def group_of(name: str) -> str:
    """Layer groups = the engine's kernels (graph/teacher.py, graph/student.py): one switch per fused op family."""
    if name.startswith("encoder.conv_stem"):
        return "stem"
    m = re.match(r"encoder\.blocks\.(\d)\.(\d)\.(conv_pwl|conv_pw|conv_dw)", name)
    if m:           # Student: expand / project convs of a stage (the SE 1x1 convs run as f32 FCs on pooled vectors: never rounded)
        return "stage%s.%s" % (m.group(1), {"conv_pw": "expand", "conv_pwl": "project", "conv_dw": "pointwise(ds)"}[m.group(3)])
    if re.match(r"encoder\.blocks\.\d\.\d\.se", name):
        return "skip"
    if name.startswith("encoder.conv1"):
        return "stem.conv1"
    if name.startswith("encoder.conv2"):
        return "stem.conv2"
    if name.startswith("encoder.layer1"):
        return "layer1.bottlenecks"
    m = re.match(r"encoder\.transition(\d)", name)
    if m:
        return "transition%s" % m.group(1)
    m = re.match(r"encoder\.stage(\d)\.(\d+)\.branches\.(\d)", name)
    if m:
        return "stage%s.branch%s.blocks" % (m.group(1), m.group(3))
    m = re.match(r"encoder\.stage(\d)\.(\d+)\.fuse_layers", name)
    if m:
        return "stage%s.fuse" % m.group(1)
    if name.startswith("encoder.incre_modules"):
        return "incre.bottlenecks"
    if name.startswith("decoder.aspp"):
        return "decoder.aspp"
    m = re.match(r"decoder\.upsampler(\d)\.conv1\.0\.conv_pw", name)
    if m:
        return "decoder.up%s.pw" % m.group(1)
    if name.startswith("decoder.upsampler2.conv2"):
        return "decoder.up2.conv2(hero)"
    if name.startswith("hm."):
        return "hm.head"
    return "other:" + name.split(".")[0] + "." + name.split(".")[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--faces", type=int, default=4)
    ap.add_argument("--out", default="")
    ap.add_argument("--budget", type=float, default=2.5e-4)
    ap.add_argument("--model", default="teacher", choices=["teacher", "student"])
    ap.add_argument("--round", default="f16", choices=["f16", "bf16", "fp8"], help="operand format of the ONE product simulated: f16 (v_mfma_f32_16x16x32_f16), "
                    "bf16, or fp8 e4m3 with a per-tensor power-of-two scale (v_mfma_f32_16x16x128_f8f6f4)")
    args = ap.parse_args()
    torch.set_num_threads(max(1, (os.cpu_count() or 2) - 1))
    weights = sw.teacher_weights() if args.model == "teacher" else sw.student_weights()
    W = ln.to_torch(weights)
    name_of = {id(t): n for n, t in W.items()}
    crops = sw.smooth_blob_images(args.faces, 256, seed=77)
    x = torch.from_numpy(crops.astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()

    real = F.conv2d
    state = {"round": set(), "seen": {}, "macs": {}}

    def rnd(t):
        if args.round == "f16":
            return t.half().float()
        if args.round == "bf16":
            return t.bfloat16().float()
        m = float(t.abs().max())                      # fp8 e4m3 (max 448): per-tensor power-of-two scale to use the range
        sc = 2.0 ** np.floor(np.log2(224.0 / m)) if m > 0 else 1.0
        return (t * sc).to(torch.float8_e4m3fn).float() / sc

    def conv(inp, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        name = name_of.get(id(w))
        if name is not None and groups == 1 and group_of(name) != "skip":
            g = group_of(name)
            state["seen"].setdefault(g, set()).add(name)
            if g in state["round"]:
                inp, w = rnd(inp), rnd(w)
        y = real(inp, w, b, stride, padding, dilation, groups)
        if name is not None and groups == 1 and group_of(name) != "skip":
            g = group_of(name)
            state["macs"].setdefault(g, {})[name] = y.shape[2] * y.shape[3] * w.shape[0] * w.shape[1] * w.shape[2] * w.shape[3]
        return y

    F.conv2d = conv
    try:
        def run(groups):
            state["round"] = set(groups)
            taps = {}
            with torch.no_grad():
                loc, score = (tn.teacher_forward if args.model == "teacher" else ln.student_forward)(W, x, taps)
            return loc.numpy(), score.numpy(), taps["hm"].numpy()

        t0 = time.time()
        loc0, score0, hm0 = run([])
        flat = hm0[:, :98].reshape(args.faces, 98, -1)
        part = np.partition(flat, -2, axis=2)
        margin = part[:, :, -1] - part[:, :, -2]
        safe = margin > 2e-3
        rng = float(np.abs(hm0).max())
        groups = sorted(state["seen"])
        macs = {g: sum(state["macs"][g].values()) for g in groups}
        total = sum(macs.values())
        print("oracle forward %.1f s; %d dense-conv groups, %.2f GMAC per face in them; heat-map range %.1f; %d / %d landmarks margin-safe" % (
            time.time() - t0, len(groups), total / 1e9, rng, int(safe.sum()), safe.size), flush=True)

        def measure(gs):
            loc, score, hm = run(gs)
            d = np.abs(loc - loc0).reshape(args.faces, 98, 2).max(2)
            moved = d > 0.5 / 64                         # picked another heat-map cell
            return {"loc_err_safe": float(d[safe & ~moved].max()) if (safe & ~moved).any() else 0.0,
                    "flips_safe": int((moved & safe).sum()), "flips_all": int(moved.sum()),
                    "hm_err": float(np.abs(hm - hm0).max()), "score_err": float(np.abs(score - score0).max())}

        table = {}
        for g in groups:
            r = measure([g])
            r["macs_share"] = macs[g] / total
            r["convs"] = len(state["seen"][g])
            table[g] = r
            print("%-28s convs %3d  MAC share %5.1f %%  loc err %.2e  hm err %.3f  flips (safe/all) %d/%d" % (
                g, r["convs"], 100 * r["macs_share"], r["loc_err_safe"], r["hm_err"], r["flips_safe"], r["flips_all"]), flush=True)
        everything = measure(groups)
        print("ALL groups on one product: loc err %.2e  hm err %.3f  flips %d/%d" % (
            everything["loc_err_safe"], everything["hm_err"], everything["flips_safe"], everything["flips_all"]), flush=True)
        # cumulative: most harmless first, keep a group if the mix stays under budget with no flip of a margin-safe landmark
        order = sorted(groups, key=lambda g: (table[g]["flips_safe"], table[g]["loc_err_safe"] / max(table[g]["macs_share"], 1e-9)))
        mix, steps = [], []
        for g in order:
            r = measure(mix + [g])
            ok = r["loc_err_safe"] < args.budget and r["flips_safe"] == 0
            steps.append({"group": g, "kept": ok, **r})
            if ok:
                mix.append(g)
            print("mix + %-28s -> loc err %.2e flips %d : %s   (mix = %.1f %% of the dense MACs)" % (
                g, r["loc_err_safe"], r["flips_safe"], "kept" if ok else "REJECTED", 100 * sum(macs[m] for m in mix) / total), flush=True)
        final = measure(mix)
        print("FINAL MIX (%d groups, %.1f %% of the dense MACs on one product): loc err %.2e, hm err %.3f, flips %d/%d" % (
            len(mix), 100 * sum(macs[m] for m in mix) / total, final["loc_err_safe"], final["hm_err"], final["flips_safe"], final["flips_all"]))
        print("three products forced for:", [g for g in groups if g not in mix])
        if args.out:
            with open(args.out, "w") as f:
                json.dump({"faces": args.faces, "budget": args.budget, "heatmap_range": rng, "margin_safe": int(safe.sum()),
                           "per_group": table, "all_on_one_product": everything, "cumulative": steps, "mix": mix, "final": final,
                           "dense_gmac_per_face": total / 1e9}, f, indent=1)
    finally:
        F.conv2d = real


if __name__ == "__main__":
    main()
